"""Fused conv epilogue kernel (csrc/epilogue.hip) against plain PyTorch fp32 math."""
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('with_res', [False, True])
@pytest.mark.parametrize('relu', [False, True])
def test_bias_act_matches_fp32_reference(dtype, with_res, relu):
    from openpifpaf_amd import fused
    torch.manual_seed(0)
    B, C, H, W = 3, 64, 37, 41
    x = torch.randn((B, C, H, W), device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn((C,), device='cuda').to(dtype)
    res = torch.randn((B, C, H, W), device='cuda').to(dtype).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    want = x.float() + bias.float().view(1, -1, 1, 1)
    if with_res:
        want = want + res.float()
    if relu:
        want = want.clamp_min(0)
    want = want.to(dtype)                     # a single rounding, like the kernel
    got = fused.bias_act_(x.clone(memory_format=torch.preserve_format), bias, res, relu)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


def test_resnet_fused_forward_equals_unfused_on_gpu():
    from openpifpaf_amd import network
    net = network.factory('resnet18').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn((2, 3, 161, 193), device='cuda').contiguous(memory_format=torch.channels_last)
    net = net.to(memory_format=torch.channels_last)
    with torch.no_grad():
        a = net(x)
        network.optimize_for_inference_(net)
        b = net(x)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) < 2e-3
