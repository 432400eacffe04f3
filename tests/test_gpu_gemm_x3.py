"""The float32 1x1 convolution on the bf16 MFMA pipe with both operands split into three bfloat16 pieces
(csrc/gemm_f32x3.hip, ``opa_gemm_bias_act_f32x3``): float32 in, float32 out, and -- the point of the test -- an error against a
float64 product that is NOT above the float32 MFMA kernel's (csrc/gemm_f32.hip) nor torch's own float32 convolution."""
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _case(cin, cout, hw, with_res, pro, seed=3):
    torch.manual_seed(seed)
    B = 3                                   # M = 3*hw*hw is not a multiple of the 128-row tile
    x = torch.randn(B, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    if not pro:
        x = x.clamp_(min=0)                 # what the kernel sees in the network: post-ReLU activations
    w = torch.randn(cout, cin, device='cuda') * (2.0 / cin) ** 0.5
    bias = torch.randn(cout, device='cuda') * 0.1
    res = torch.randn(B, cout, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last) if with_res else None
    a_bias = torch.randn(cin, device='cuda') * 0.3 if pro else None
    xa = x.double()
    if pro:
        xa = (xa + a_bias.double().view(1, -1, 1, 1)).clamp_(min=0)
    ref = torch.nn.functional.conv2d(xa, w.double().view(cout, cin, 1, 1), bias.double())
    if with_res:
        ref = ref + res.double()
    return x, w, bias, res, a_bias, ref.clamp_(min=0)


def _rms(out, ref):
    d = out.double() - ref
    return float((d * d).mean().sqrt()) / float(ref.abs().max())


@pytest.mark.parametrize('cin,cout,hw,with_res,pro', [(1024, 256, 27, False, False), (256, 1024, 27, True, False),
                                                      (2048, 512, 13, False, False), (512, 2048, 13, True, True),
                                                      (64, 64, 23, False, False), (128, 512, 19, True, True)])
@pytest.mark.parametrize('terms', [6, 9])
def test_split_operand_gemm_is_at_least_as_exact_as_the_float32_mfma(cin, cout, hw, with_res, pro, terms):
    from openpifpaf_amd import fused
    x, w, bias, res, a_bias, ref = _case(cin, cout, hw, with_res, pro)
    w3 = fused.split_weight(w)
    assert w3.dtype == torch.bfloat16 and tuple(w3.shape) == (3, cout, cin)
    assert torch.equal(w3.float().sum(0), w) and torch.equal((w3[0].float() + w3[1].float()) + w3[2].float(), w)
    got = fused.conv1x1_bias_act_x3(x, w3, bias, res, True, a_bias, terms)
    native = fused.conv1x1_bias_act(x, w, bias, res, True, a_bias)
    assert got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
    e_got, e_native = _rms(got, ref), _rms(native, ref)
    assert e_got <= 1.05 * e_native + 1e-9, (e_got, e_native)        # measured: 0.35-0.75 of the float32 MFMA's error
    assert e_got < 2e-7, e_got
    assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


def test_split_operand_gemm_rejects_what_it_cannot_run():
    import ctypes
    from openpifpaf_amd import _lib, fused
    x, w, bias, _, _, _ = _case(96, 64, 9, False, False)             # K = 96 is not a multiple of 64
    w3 = fused.split_weight(w)
    out = torch.empty((3, 64, 9, 9), device='cuda').contiguous(memory_format=torch.channels_last)
    rc = _lib.lib().opa_gemm_bias_act_f32x3(ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(w3.data_ptr()),
                                            ctypes.c_void_p(bias.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), 243, 64, 96, 1, 6,
                                            None)
    assert rc != 0
    x, w, bias, _, _, _ = _case(64, 64, 9, False, False)
    rc = _lib.lib().opa_gemm_bias_act_f32x3(ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(fused.split_weight(w).data_ptr()),
                                            ctypes.c_void_p(bias.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), 243, 64, 64, 1, 7,
                                            None)
    assert rc != 0                                                  # terms is 6 or 9


@pytest.mark.parametrize('k1,k2,n,hw_in,stride,pro', [(64, 64, 256, 37, 1, False), (128, 256, 512, 41, 2, True), (256, 512, 1024, 27, 2, True),
                                                      (512, 1024, 2048, 14, 2, False), (64, 192, 128, 9, 3, True)])
def test_block_tail_and_downsample_as_one_product(k1, k2, n, hw_in, stride, pro):
    """``relu(conv3(h) + downsample(x) + bias)`` as ONE product of the split-operand kernel (``opa_gemm2_bias_act_f32x3``) against
    float64, and against the two-launch form (downsampling convolution by torch, then the GEMM with a residual)."""
    from openpifpaf_amd import fused
    torch.manual_seed(11)
    B = 3
    ho = (hw_in - 1) // stride + 1
    x = torch.randn(B, k2, hw_in, hw_in + 2, device='cuda').clamp_(min=0).contiguous(memory_format=torch.channels_last)
    wo = (hw_in + 2 - 1) // stride + 1
    h = torch.randn(B, k1, ho, wo, device='cuda').contiguous(memory_format=torch.channels_last)
    if not pro:
        h = h.clamp_(min=0)
    conv = torch.nn.Conv2d(k1, n, 1, bias=False).cuda()
    dconv = torch.nn.Conv2d(k2, n, 1, stride, bias=False).cuda()
    bias = torch.randn(n, device='cuda') * 0.1
    a_bias = torch.randn(k1, device='cuda') * 0.3 if pro else None
    with torch.no_grad():
        assert fused.pair_supported(conv, dconv, h, x, bias, a_bias)
        got = fused.conv1x1_pair_bias_act_x3(conv, dconv, h, x, bias, True, a_bias)
        hd = h.double()
        if pro:
            hd = (hd + a_bias.double().view(1, -1, 1, 1)).clamp_(min=0)
        ref = (torch.nn.functional.conv2d(hd, conv.weight.double()) + torch.nn.functional.conv2d(x.double(), dconv.weight.double(), stride=stride)
               + bias.double().view(1, -1, 1, 1)).clamp_(min=0)
        two = fused.conv1x1_bias_act(h, conv.weight.reshape(n, k1), bias, dconv(x).contiguous(memory_format=torch.channels_last), True, a_bias)
    assert tuple(got.shape) == (B, n, ho, wo) and got.is_contiguous(memory_format=torch.channels_last)
    e_got, e_two = _rms(got, ref), _rms(two, ref)
    assert e_got <= 1.05 * e_two + 1e-9 and e_got < 2e-7, (e_got, e_two)
    assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


def test_resnet50_trunk_with_and_without_the_split_operand_kernel():
    """The whole float32 trunk: the heads agree within 1e-4 of their largest magnitude whichever kernel the 1x1 convolutions take
    (the bar the Winograd kernel was held to), and the 'gemm3' choice is really taken."""
    from openpifpaf_amd import fused, network
    torch.manual_seed(5)
    net = network.factory('resnet50').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    network.optimize_for_inference_(net)
    net = net.to(memory_format=torch.channels_last)
    x = torch.randn((2, 3, 193, 161), device='cuda').contiguous(memory_format=torch.channels_last)
    saved, terms, pair, conv3, stem, force = fused.choices(), fused.X3_TERMS, fused.X3_PAIR, fused.X3_CONV3, fused.X3_STEM, fused.FORCE_PICK
    try:
        fused.X3_TERMS = 6
        fused.X3_PAIR = fused.X3_CONV3 = fused.X3_STEM = False
        fused.set_choices({k: 'gemm' for k in saved}, replace=True)
        with torch.no_grad():
            fused.set_choices({}, replace=True)
            import os
            os.environ['OPA_CONV1X1'] = 'gemm'
            a = net(x)
            forced = {k: 'gemm3' for k, v in fused.choices().items() if k[0] == 'torch.float32' and k[2] % 64 == 0}
            assert forced
            fused.set_choices(forced)
            fused.FORCE_PICK = 'x3'
            fused.X3_PAIR = fused.X3_CONV3 = fused.X3_STEM = True                     # ... the blocks' tails with their downsampling convolutions, the strided 3x3
            b = net(x)
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 1e-4 * float(u.abs().max()), float((u - v).abs().max())
        assert not all(torch.equal(u, v) for u, v in zip(a, b))      # (another kernel did run)
    finally:
        os.environ.pop('OPA_CONV1X1', None)
        fused.X3_TERMS, fused.X3_PAIR, fused.X3_CONV3, fused.X3_STEM, fused.FORCE_PICK = terms, pair, conv3, stem, force
        fused.set_choices(saved, replace=True)


@pytest.mark.parametrize('c,n,h,w,stride', [(128, 128, 37, 41, 2), (256, 256, 20, 27, 2), (64, 128, 17, 9, 1), (512, 512, 11, 12, 2),
                                            (64, 64, 9, 30, 3)])
def test_strided_3x3_convolution_as_implicit_gemm(c, n, h, w, stride):
    """``opa_conv3x3_f32x3``: padding 1, any stride, against a float64 convolution and against torch's float32 one."""
    from openpifpaf_amd import fused
    torch.manual_seed(13)
    x = torch.randn(3, c, h, w, device='cuda').clamp_(min=0).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(c, n, 3, stride, 1, bias=False).cuda()
    bias = torch.randn(n, device='cuda') * 0.1
    with torch.no_grad():
        assert fused.conv3x3_x3_supported(conv, x, bias)
        got = fused.conv3x3_bias_act_x3(conv, x, bias, True)
        ref = (torch.nn.functional.conv2d(x.double(), conv.weight.double(), stride=stride, padding=1) + bias.double().view(1, -1, 1, 1)).clamp_(min=0)
        theirs = torch.relu(conv(x) + bias.view(1, -1, 1, 1))
    assert tuple(got.shape) == tuple(ref.shape) and got.is_contiguous(memory_format=torch.channels_last)
    e_got, e_theirs = _rms(got, ref), _rms(theirs, ref)
    # (measured: 0.4-1.4 x the error of whatever algorithm MIOpen picks for the shape; float32-grade either way)
    assert e_got <= 2.0 * e_theirs + 1e-9 and e_got < 1e-7, (e_got, e_theirs)
    assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


@pytest.mark.parametrize('h,w,n', [(641, 641, 64), (321, 193, 64), (37, 52, 128)])
def test_resnet_stem_as_implicit_gemm(h, w, n):
    """``opa_conv_rows_f32x3`` through ``fused.stem7x7_bias_act_x3``: the 7x7 stride-2 padding-3 convolution on RGB input."""
    from openpifpaf_amd import fused
    torch.manual_seed(17)
    x = torch.randn(2, 3, h, w, device='cuda')
    conv = torch.nn.Conv2d(3, n, 7, 2, 3, bias=False).cuda()
    bias = torch.randn(n, device='cuda') * 0.1
    with torch.no_grad():
        for xin in (x, x.contiguous(memory_format=torch.channels_last)):
            assert fused.stem_x3_supported(conv, xin, bias)
            got = fused.stem7x7_bias_act_x3(conv, xin, bias, True)
            ref = (torch.nn.functional.conv2d(x.double(), conv.weight.double(), stride=2, padding=3) + bias.double().view(1, -1, 1, 1)).clamp_(min=0)
            theirs = torch.relu(conv(xin) + bias.view(1, -1, 1, 1))
            assert tuple(got.shape) == tuple(ref.shape) and got.is_contiguous(memory_format=torch.channels_last)
            e_got, e_theirs = _rms(got, ref), _rms(theirs, ref)
            assert e_got <= 2.0 * e_theirs + 1e-9 and e_got < 1e-7, (e_got, e_theirs)
            assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


def test_head_convolution_with_padded_output_channels():
    """``fused.head_conv_x3``: a biased 1x1 convolution whose output channels (340 = 17 x 5 x 4) are no multiple of 64."""
    from openpifpaf_amd import fused
    torch.manual_seed(19)
    x = torch.randn(2, 2048, 21, 23, device='cuda').clamp_(min=0).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(2048, 340, 1).cuda()
    with torch.no_grad():
        assert fused.head_conv_x3_supported(conv, x)
        got = fused.head_conv_x3(conv, x)
        ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double())
        theirs = conv(x)
    assert tuple(got.shape) == (2, 340, 21, 23) and got.is_contiguous(memory_format=torch.channels_last)
    assert _rms(got, ref) <= 1.05 * _rms(theirs, ref) + 1e-9 and _rms(got, ref) < 1e-7
