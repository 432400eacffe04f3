"""1x1 GEMM with the operand prologue (bias+ReLU of the preceding 3x3 conv) vs. separate epilogue pass + GEMM."""
import sys, torch
sys.path.insert(0, '.')
from openpifpaf_amd import fused
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
B = 32
for (cin, cout, hw) in [(64, 256, 321), (128, 512, 161), (256, 1024, 81), (512, 2048, 41)]:
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, device='cuda') / cin ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda').to(torch.bfloat16)
    ab = torch.randn(cin, device='cuda').to(torch.bfloat16)
    r = torch.randn(B, cout, hw, hw, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_plain = bench(lambda: fused.conv1x1_bias_act(x, w, b, r, True))
    t_pro = bench(lambda: fused.conv1x1_bias_act(x, w, b, r, True, a_bias=ab))
    t_pass = bench(lambda: fused.bias_act_(x, ab))
    print('K=%4d N=%4d hw=%3d: GEMM %.3f ms, GEMM+prologue %.3f ms, separate bias_act pass %.3f ms -> saves %.3f ms' % (
        cin, cout, hw, t_plain, t_pro, t_pass, t_plain + t_pass - t_pro), flush=True)
