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


@pytest.mark.parametrize('cin,cout,hw,with_res', [(64, 256, 37, True), (256, 64, 41, False), (128, 512, 19, True),
                                                  (2048, 512, 9, False), (64, 64, 23, False)])
def test_fused_conv1x1_gemm_matches_fp32_reference(cin, cout, hw, with_res):
    from openpifpaf_amd import fused
    torch.manual_seed(1)
    B = 3                                   # M = 3*hw*hw is not a multiple of the 128-row tile
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w4 = (torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda').to(torch.bfloat16)
    r = torch.randn(B, cout, hw, hw, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    ref = torch.nn.functional.conv2d(x.float(), w4.float(), b.float())
    if with_res:
        ref = ref + r.float()
    for relu in (True, False):
        want = ref.clamp_min(0) if relu else ref
        got = fused.conv1x1_bias_act(x, w4.reshape(cout, cin).contiguous(), b, r, relu)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        err = (got.float() - want).abs()
        tol = 2.0 ** -7 * want.abs().clamp_min(1.0)          # bf16 output rounding (8 bits of mantissa)
        assert bool((err <= tol).all()), float(err.max())


@pytest.mark.parametrize('cin,cout,hw,with_res', [(64, 256, 37, True), (128, 512, 19, True), (512, 2048, 9, False)])
def test_gemm_with_fused_operand_prologue_equals_two_passes(cin, cout, hw, with_res):
    """opa_gemm_pro_bias_act_bf16: relu(x + a_bias) applied while the A tile is staged must give the very
    same bits as the separate bias_act pass followed by the plain fused GEMM."""
    from openpifpaf_amd import fused
    torch.manual_seed(2)
    B = 3
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.7).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ab = (torch.randn(cin, device='cuda') * 0.3).to(torch.bfloat16)
    w = (torch.randn(cout, cin, device='cuda') / cin ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda').to(torch.bfloat16)
    r = torch.randn(B, cout, hw, hw, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    want = fused.conv1x1_bias_act(fused.bias_act_(x.clone(), ab), w, b, r, True)
    got = fused.conv1x1_bias_act(x, w, b, r, True, a_bias=ab)
    assert torch.equal(got, want)


@pytest.mark.parametrize('cin,cout,hw,with_res,pro', [(64, 256, 37, True, False), (64, 256, 37, True, True), (256, 64, 41, False, False),
                                                      (2048, 512, 9, False, False), (96, 192, 23, True, True), (32, 64, 5, False, False)])
def test_float32_conv1x1_gemm_matches_the_convolution(cin, cout, hw, with_res, pro):
    """opa_gemm_bias_act_f32 (v_mfma_f32_32x32x2f32, f32 operands and accumulation): the float32 1x1 convolution with
    bias / residual / ReLU -- and optionally the preceding convolution's bias + ReLU on its operand -- against
    PyTorch's float32 convolution.  Tolerance: a float32 dot product of `cin` terms summed in another order."""
    from openpifpaf_amd import fused
    torch.manual_seed(4)
    B = 3                                   # M = 3*hw*hw is not a multiple of the 128-row tile
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.5).contiguous(memory_format=torch.channels_last)
    w4 = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    b = torch.randn(cout, device='cuda')
    ab = torch.randn(cin, device='cuda') * 0.3 if pro else None
    r = torch.randn(B, cout, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last) if with_res else None
    assert fused.conv1x1_supported(x, w4, b, r, ab)
    xin = (x + ab.view(1, -1, 1, 1)).clamp_min(0) if pro else x
    ref = torch.nn.functional.conv2d(xin.double(), w4.double(), b.double())
    if with_res:
        ref = ref + r.double()
    for relu in (True, False):
        want = ref.clamp_min(0) if relu else ref
        got = fused.conv1x1_bias_act(x, w4.reshape(cout, cin).contiguous(), b, r, relu, ab)
        assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        err = float((got.double() - want).abs().max())
        assert err <= 2e-6 * cin ** 0.5 * float(want.abs().max().clamp_min(1.0)), err
    # not for these: another dtype among the operands, K not a multiple of 32
    assert not fused.conv1x1_supported(x, w4.to(torch.bfloat16), b, r, ab)
    assert not fused.conv1x1_supported(x[:, :cin - 8].contiguous(memory_format=torch.channels_last), w4[:, :cin - 8], b)


def test_resnet50_float32_fused_gemm_forward_equals_unfused():
    """The float32 network with the fused 1x1 GEMMs forced on against the unfused PyTorch network."""
    import os
    from openpifpaf_amd import fused, network
    net = network.factory('resnet50').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn((2, 3, 161, 193), device='cuda')
    old, saved = os.environ.get('OPA_CONV1X1'), fused.choices()
    try:
        os.environ['OPA_CONV1X1'] = 'gemm'
        with torch.no_grad():
            want = net(x)
            network.optimize_for_inference_(net)
            net = net.to(memory_format=torch.channels_last)
            got = net(x.contiguous(memory_format=torch.channels_last))
        assert any(k[0] == 'torch.float32' and v == 'gemm' for k, v in fused.choices().items())
    finally:
        if old is None:
            os.environ.pop('OPA_CONV1X1', None)
        else:
            os.environ['OPA_CONV1X1'] = old
        fused._CHOICE.clear()
        fused.set_choices(saved)
    for u, v in zip(got, want):
        assert u.shape == v.shape and u.dtype == torch.float32
        assert float((u - v).abs().max()) <= 2e-3 * float(v.abs().max().clamp_min(1.0))


def test_resnet50_fused_gemm_forward_close_to_unfused():
    from openpifpaf_amd import network
    net = network.factory('resnet50').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn((2, 3, 161, 193), device='cuda')
    with torch.no_grad():
        want = net(x)
        network.optimize_for_inference_(net)
        net = net.to(memory_format=torch.channels_last).to(torch.bfloat16)
        got = net(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    for u, v in zip(got, want):
        assert u.shape == v.shape and u.dtype == torch.float32
        # bf16 network vs fp32 network: only a sanity bound (confidences are in [0,1])
        assert float((u[:, :, 1] - v[:, :, 1]).abs().mean()) < 0.05


def test_fused_gemm_refuses_mixed_precision_operands():
    """Autocast keeps parameters in float32: bf16 activations with fp32 weight / bias / prologue bias, or a
    residual of another layout, must take the PyTorch path -- the kernel would reinterpret the raw buffers
    (ADVICE r1).  The result has to match the fp32 reference either way."""
    from openpifpaf_amd import fused
    torch.manual_seed(3)
    B, cin, cout, hw = 2, 64, 128, 17
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()             # float32 parameters
    bias32 = torch.randn(cout, device='cuda')
    w16, b16 = conv.weight.detach().to(torch.bfloat16), bias32.to(torch.bfloat16)
    res_nchw = torch.randn(B, cout, hw, hw, device='cuda').to(torch.bfloat16)       # NOT channels_last
    assert fused.conv1x1_supported(x, w16, b16)
    assert not fused.conv1x1_supported(x, conv.weight, b16)
    assert not fused.conv1x1_supported(x, w16, bias32)
    assert not fused.conv1x1_supported(x, w16, b16, a_bias=torch.zeros(cin, device='cuda'))
    assert not fused.conv1x1_supported(x, w16, b16, residual=res_nchw)
    assert not fused.conv1x1_supported(x, w16, b16, residual=res_nchw[:, :64].contiguous(memory_format=torch.channels_last))
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        got = fused.conv_bias_act(conv, x, bias32.to(torch.bfloat16), None, True)   # fp32 weight under autocast
    want = torch.nn.functional.conv2d(x.float(), conv.weight.float(), bias32.to(torch.bfloat16).float()).clamp_min(0)
    assert float((got.float() - want).abs().max()) <= 2.0 ** -6 * float(want.abs().max().clamp_min(1.0))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('hw', [(41, 41), (9, 13)])
def test_fused_head_epilogue_equals_the_composite_field(dtype, hw):
    """opa_head_epilogue against the PyTorch restatement of CompositeField4's post-processing (itself pinned to the
    reference's class in test_oracle_vs_reference.py): PixelShuffle, crop, layout, sigmoid, offsets, softplus.
    Index offsets and the layout are exact; sigmoid / softplus may differ from ATen's device functions in the
    last place (tolerance 2 ulp of float32)."""
    from openpifpaf_amd import headmeta, network
    torch.manual_seed(5)
    for meta in headmeta.cocokp_metas():
        head = network.CompositeField4(meta, 64).cuda().eval().to(memory_format=torch.channels_last)
        if dtype != torch.float32:
            head = head.to(dtype)
        feat = (torch.randn(2, 64, hw[0], hw[1], device='cuda') * 3).to(dtype).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            head.fused_epilogue = False
            want = head(feat)
            head.fused_epilogue = True
            got = head(feat)
        assert got.shape == want.shape == (2, meta.n_fields, head.n_components, 2 * hw[0] - 1, 2 * hw[1] - 1)
        assert got.dtype == torch.float32
        assert torch.equal(got[:, :, 0], want[:, :, 0])                   # raw component: layout only
        ulp = torch.finfo(torch.float32).eps * want.abs().clamp_min(1e-30)
        assert bool(((got - want).abs() <= 2 * ulp).all()), float(((got - want).abs() / ulp).max())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('k,stride,C,hw', [(5, 1, 174, (41, 37)), (5, 2, 24, (33, 40)), (3, 1, 64, (9, 9)), (5, 2, 348, (21, 21))])
def test_depthwise_conv_kernel_matches_fp32_reference(dtype, k, stride, C, hw):
    """opa_dwconv_bias_act against torch's depthwise conv2d in float32, on a dense tensor and on a channel slice
    (the second half of a wider channels-last tensor, as the ShuffleNet unit feeds it)."""
    from openpifpaf_amd import fused
    torch.manual_seed(11)
    wide = (torch.randn(2, 2 * C, hw[0], hw[1], device='cuda')).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, 1, k, k, device='cuda') / k).to(dtype)
    b = torch.randn(C, device='cuda').to(dtype)
    w_taps = w.reshape(C, k * k).t().contiguous()
    for x in (wide[:, C:], wide[:, :C].contiguous(memory_format=torch.channels_last)):
        assert fused.dwconv_supported(x, k, stride)
        want = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=stride, padding=k // 2, groups=C)
        for relu in (False, True):
            got = fused.dwconv_bias_act(x, w_taps, b, k, stride, relu)
            ref = want.clamp_min(0) if relu else want
            assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
            tol = (2.0 ** -7 if dtype == torch.bfloat16 else 1e-5) * ref.abs().clamp_min(1.0)
            assert bool(((got.float() - ref).abs() <= tol).all()), float((got.float() - ref).abs().max())


def test_channel_interleave_equals_cat_and_shuffle():
    from openpifpaf_amd import fused, network
    torch.manual_seed(12)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(2, 348, 13, 17, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
        a, _ = x.chunk(2, dim=1)
        b = torch.randn(2, 174, 13, 17, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
        want = network._channel_shuffle(torch.cat((a, b), dim=1), 2)
        got = fused.channel_interleave(a, b)
        assert torch.equal(got, want) and got.is_contiguous(memory_format=torch.channels_last)


def test_shufflenet_fused_forward_close_to_unfused():
    from openpifpaf_amd import network
    net = network.factory('shufflenetv2k16').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn((2, 3, 161, 193), device='cuda')
    with torch.no_grad():
        want = net(x)
        network.optimize_for_inference_(net)
        net = net.to(memory_format=torch.channels_last)
        got = net(x.contiguous(memory_format=torch.channels_last))
    for u, v in zip(got, want):
        assert u.shape == v.shape
        assert float((u - v).abs().max()) < 5e-3, float((u - v).abs().max())
