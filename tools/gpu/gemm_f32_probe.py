"""float32 1x1 convolution: the fused MFMA GEMM against MIOpen's convolution + the fused epilogue pass,
on the bottleneck shapes of the bench network (resnet50, 641 px, batch 32)."""
import sys
import torch
sys.path.insert(0, '.')
from openpifpaf_amd import fused

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')


def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps


print('shape                         | conv+epilogue ms | fused gemm ms | TFLOP/s | GB/s  | max |delta| / max |ref|')
for name, hw, cin, cout, res, pro in [
        ('layer1 conv3 64->256 +res', 321, 64, 256, True, False), ('layer1 conv3 (pro)', 321, 64, 256, True, True),
        ('layer1 conv1 256->64', 321, 256, 64, False, False),
        ('layer2 conv3 128->512 +res', 161, 128, 512, True, False), ('layer2 conv1 512->128', 161, 512, 128, False, False),
        ('layer3 conv3 256->1024 +res', 81, 256, 1024, True, False), ('layer3 conv1 1024->256', 81, 1024, 256, False, False),
        ('layer4 conv3 512->2048 +res', 41, 512, 2048, True, False), ('layer4 conv1 2048->512', 41, 2048, 512, False, False)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn((B, cin, hw, hw), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bias = torch.randn((cout,), device=dev, generator=g)
    a_bias = torch.randn((cin,), device=dev, generator=g) if pro else None
    r = torch.randn((B, cout, hw, hw), device=dev, generator=g).contiguous(memory_format=torch.channels_last) if res else None
    w2d = conv.weight.detach().reshape(cout, cin).contiguous()
    assert fused.conv1x1_supported(x, conv.weight, bias, r, a_bias), name
    with torch.no_grad():
        xin = torch.relu(x + a_bias.view(1, -1, 1, 1)) if pro else x
        ref = conv(xin) + bias.view(1, -1, 1, 1)
        if res:
            ref = ref + r
        ref = torch.relu(ref)
        got = fused.conv1x1_bias_act(x, w2d, bias, r, True, a_bias)
        err = float((got - ref).abs().max() / ref.abs().max())
        t_conv = t_ms(lambda: fused.bias_act_(conv(xin), bias, r, True))
        t_gemm = t_ms(lambda: fused.conv1x1_bias_act(x, w2d, bias, r, True, a_bias))
    M = B * hw * hw
    flops = 2.0 * M * cin * cout
    byts = 4.0 * (M * cin + M * cout * (2 if res else 1) + cin * cout)
    print('%-29s | %16.3f | %13.3f | %7.1f | %5.0f | %.2e' % (name, t_conv, t_gemm, flops / t_gemm / 1e9, byts / t_gemm / 1e6, err))
    del x, r, ref, got
    torch.cuda.empty_cache()
