"""csrc/winograd.hip on the GPU: F(2x2, 3x3) against a float64 convolution (tolerance: 2e-5 of the output's largest
magnitude; torch's own float32 convolution is printed beside it by tools/gpu/winograd_probe.py), every variant / workgroup
order, ragged sizes (odd H / W, a last tile block of one tile), bias + ReLU, and the ResNet trunk with and without it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, C, O, H, W, variant, order, bias=False, relu=False, seed=0):
    from openpifpaf_amd import winograd
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn((O, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn((O,), generator=g).cuda() if bias else None
    y = winograd.conv3x3(x, winograd.transform_filter(w, variant), O, bias=b, relu=relu, variant=variant, order=order)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    if relu:
        ref = ref.relu()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    return (y.double() - ref).abs().max().item() / ref.abs().max().item()


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', [(1, 16, 64, 8, 8), (2, 32, 64, 7, 9), (3, 64, 128, 21, 21), (2, 64, 64, 41, 40),
                                   (1, 128, 128, 5, 3), (1, 16, 64, 1, 1), (5, 48, 192, 16, 13)])
def test_winograd_equals_the_convolution(variant, shape):
    for order in (0, 1):
        assert _case(*shape, variant, order) < 2e-5


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
def test_bias_and_relu_in_the_output_transform(variant):
    assert _case(2, 32, 64, 11, 13, variant, 0, bias=True, relu=True) < 2e-5
    assert _case(2, 32, 64, 11, 13, variant, 1, bias=True, relu=False) < 2e-5


def test_bad_arguments_are_refused():
    from openpifpaf_amd import winograd
    x = torch.randn((1, 24, 8, 8)).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn((64, 24, 3, 3)).cuda()
    assert not winograd.supported(x, w, 0) and winograd.supported(x, w, 1)
    with pytest.raises(Exception):
        winograd.conv3x3(x, torch.zeros(16 * 24 * 64, device='cuda'), 64, variant=0)


def test_resnet_fields_with_and_without_winograd():
    """The float32 trunk through the Winograd kernel against the same trunk through MIOpen's convolutions: head outputs
    within 1e-4 of the largest magnitude (review of round 5, item 8)."""
    from openpifpaf_amd import network, winograd
    net = network.optimize_for_inference_(network.factory('resnet50')).cuda().to(memory_format=torch.channels_last)
    x = torch.randn((2, 3, 321, 321), generator=torch.Generator().manual_seed(1)).cuda().contiguous(memory_format=torch.channels_last)
    old = winograd.set_mode('winograd')
    try:
        with torch.no_grad():
            a = net(x)
            winograd.set_mode('conv')
            b = net(x)
    finally:
        winograd.set_mode(old)
    for fa, fb in zip(a, b):
        raw_a, raw_b = torch.nan_to_num(fa), torch.nan_to_num(fb)
        assert not torch.equal(raw_a, raw_b)                  # (it did take the other path)
        assert (raw_a - raw_b).abs().max().item() <= 1e-4 * raw_b.abs().max().item()


def test_basic_block_networks_take_the_kernel_too():
    """resnet18 (BasicBlock: two 3x3 convolutions per block) at a batch that fills the chip: fields within 1e-4 of the MIOpen trunk."""
    from openpifpaf_amd import network, winograd
    net = network.optimize_for_inference_(network.factory('resnet18')).cuda().to(memory_format=torch.channels_last)
    assert sum(1 for m in net.modules() if hasattr(m, 'wino_u1')) == 5 and sum(1 for m in net.modules() if hasattr(m, 'wino_u2')) == 8
    x = torch.randn((16, 3, 321, 321), generator=torch.Generator().manual_seed(2)).cuda().contiguous(memory_format=torch.channels_last)
    old = winograd.set_mode('winograd')
    try:
        with torch.no_grad():
            a = net(x)
            winograd.set_mode('conv')
            b = net(x)
    finally:
        winograd.set_mode(old)
    for fa, fb in zip(a, b):
        raw_a, raw_b = torch.nan_to_num(fa), torch.nan_to_num(fb)
        assert not torch.equal(raw_a, raw_b)
        assert (raw_a - raw_b).abs().max().item() <= 1e-4 * raw_b.abs().max().item()
