"""Correctness + speed of the fused 1x1-conv GEMM vs conv2d + bias_act_."""
import sys, time, torch
sys.path.insert(0, '.')
from openpifpaf_amd import fused
torch.backends.cudnn.benchmark = True
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
torch.manual_seed(0)
B = 32
for (cin, cout, hw, res) in [(64, 256, 321, True), (256, 64, 321, False), (64, 64, 321, False), (128, 512, 161, True), (512, 128, 161, False),
                             (256, 1024, 81, True), (1024, 256, 81, False), (512, 2048, 41, True), (2048, 512, 41, False)]:
    x = (torch.randn(B, cin, hw, hw, device='cuda') * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w4 = (torch.randn(cout, cin, 1, 1, device='cuda') * (1.0 / cin ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w2 = w4.reshape(cout, cin).contiguous()
    b = torch.randn(cout, device='cuda').to(torch.bfloat16)
    r = torch.randn(B, cout, hw, hw, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    ref = torch.nn.functional.conv2d(x.float(), w4.float(), b.float())
    if res: ref = ref + r.float()
    ref = ref.clamp_min(0)
    got = fused.conv1x1_bias_act(x, w2, b, r, True)
    err = (got.float() - ref).abs().max().item(); scale = ref.abs().max().item()
    t_mine = bench(lambda: fused.conv1x1_bias_act(x, w2, b, r, True))
    t_ref = bench(lambda: fused.bias_act_(torch.nn.functional.conv2d(x, w4, None), b, r, True))
    M = B * hw * hw
    gb = (M * cin + M * cout * (2 if res else 1)) * 2 / 1e9
    print('K=%4d N=%4d hw=%3d res=%-5s err %.3g (max %.1f)  fused GEMM %.3f ms (%.2f TB/s, %.0f TF/s)  conv+bias_act %.3f ms  -> %.2fx' % (
        cin, cout, hw, res, err, scale, t_mine, gb / t_mine, 2.0 * M * cin * cout / t_mine / 1e9, t_ref, t_ref / t_mine), flush=True)
