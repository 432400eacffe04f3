"""Winograd F(2x2, 3x3) host side (openpifpaf_amd/winograd.py): the transform matrices, the filter operand's layout as the
kernel's lanes index it (csrc/winograd.hip: u_ptr), the shape rule.  No GPU."""
import numpy as np
import pytest
import torch

from openpifpaf_amd import network, winograd


@pytest.mark.parametrize('shape', [(1, 8, 4, 5, 7), (2, 3, 5, 6, 6), (1, 4, 4, 1, 1), (2, 16, 8, 9, 4)])
def test_reference_model_equals_the_direct_convolution(shape):
    B, C, O, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + H)
    x, w = torch.randn((B, C, H, W), generator=g), torch.randn((O, C, 3, 3), generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert (winograd.reference_f23(x, w) - ref).abs().max().item() < 1e-12


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
def test_filter_operand_layout_is_what_the_lanes_read(variant):
    """Lane l of the wave that owns position p reads float4 number ((((block * chunks + chunk) * 16 + p) * NB + j) * KQ + kq)
    * 64 + l and multiplies its e-th element as B[k = l // 32][c = l % 32] of MFMA e: that must be U[p][chunk * KC + 2 * (4 *
    kq + e) + l // 32][32 * (block * NB + j) + l % 32]."""
    kc, nb = winograd.VARIANTS[variant]
    cin, cout = 2 * kc, 64 * nb
    g = torch.Generator().manual_seed(3)
    w = torch.randn((cout, cin, 3, 3), generator=g)
    u = winograd.transform_filter(w, variant).numpy().reshape(-1, 4)
    G = np.array(winograd._G)
    U = np.einsum('ar,oirs,bs->aboi', G, w.double().numpy(), G).reshape(16, cout, cin)
    kq_n, chunks = kc // 8, cin // kc
    rng = np.random.default_rng(0)
    for _ in range(400):
        block, chunk, p, j, kq, lane, e = (rng.integers(cout // (32 * nb)), rng.integers(chunks), rng.integers(16),
                                           rng.integers(nb), rng.integers(kq_n), rng.integers(64), rng.integers(4))
        idx = ((((block * chunks + chunk) * 16 + p) * nb + j) * kq_n + kq) * 64 + lane
        k = chunk * kc + 2 * (4 * kq + e) + lane // 32
        c = 32 * (block * nb + j) + lane % 32
        assert u[idx, e] == np.float32(U[p, c, k])
    assert u.shape[0] * 4 == 16 * cin * cout


def test_shape_rule_and_fallback_on_the_cpu():
    assert winograd.workgroups((32, 64, 321, 321), 64) == 12961
    assert winograd.workgroups((1, 512, 41, 41), 512) == 7 * 8
    conv = torch.nn.Conv2d(16, 64, 3, 1, 1, bias=False)
    x = torch.randn(1, 16, 6, 6)
    u = winograd.transform_filter(conv.weight)
    with torch.no_grad():                                         # CPU tensor: torch's convolution
        assert torch.equal(winograd.conv_or_fallback(conv, x, u), conv(x))
    assert not winograd.supported(x, conv.weight)


def test_bottlenecks_carry_the_transformed_filter_of_their_stride_one_convolutions():
    net = network.optimize_for_inference_(network.factory('resnet50'))
    blocks = [m for m in net.modules() if isinstance(m, network._Bottleneck)]
    assert len(blocks) == 16
    with_u = [m for m in blocks if hasattr(m, 'wino_u')]
    assert len(with_u) == 13 and all(m.conv2.stride == (1, 1) for m in with_u)
    assert all(m.wino_u.numel() == 16 * m.conv2.in_channels * m.conv2.out_channels for m in with_u)
    assert not any(k.endswith('wino_u') for k in net.state_dict())        # derived data: not part of a checkpoint
    net.to(torch.bfloat16)                                                 # a bfloat16 network keeps MIOpen's convolution
    assert with_u[0].wino_u.dtype == torch.bfloat16


def test_mode_switch():
    old = winograd.set_mode('conv')
    try:
        assert winograd.get_mode() == 'conv'
        conv = torch.nn.Conv2d(16, 64, 3, 1, 1, bias=False)
        assert not winograd.takes(conv, torch.randn(1, 16, 6, 6), winograd.transform_filter(conv.weight))
        with pytest.raises(ValueError):
            winograd.set_mode('fast')
    finally:
        winograd.set_mode(old)
    assert winograd.get_mode() == old
