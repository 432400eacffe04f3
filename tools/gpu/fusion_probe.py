"""Probe: does torch.miopen_convolution_relu fuse bias+ReLU into the conv for bf16/fp16 NHWC?"""
import sys, time, torch
sys.path.insert(0, '.')
from openpifpaf_amd import fused
torch.backends.cudnn.benchmark = True
dev = 'cuda'
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for dtype in (torch.bfloat16, torch.float16):
    for (cin, cout, k, hw) in [(256, 64, 1, 321), (64, 64, 3, 321), (64, 256, 1, 321), (512, 128, 1, 161)]:
        x = torch.randn(32, cin, hw, hw, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, k, k, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last) * 0.05
        b = torch.randn(cout, device=dev, dtype=dtype)
        pad = k // 2
        t_conv = bench(lambda: torch.nn.functional.conv2d(x, w, None, 1, pad))
        t_mine = bench(lambda: fused.bias_act_(torch.nn.functional.conv2d(x, w, None, 1, pad), b))
        try:
            y = torch.miopen_convolution_relu(x, w, b, [1, 1], [pad, pad], [1, 1], 1)
            ref = torch.relu(torch.nn.functional.conv2d(x, w, b, 1, pad))
            err = (y.float() - ref.float()).abs().max().item()
            t_fused = bench(lambda: torch.miopen_convolution_relu(x, w, b, [1, 1], [pad, pad], [1, 1], 1))
            msg = 'miopen_convolution_relu %.3f ms (err %.3g, channels_last out=%s)' % (t_fused, err, y.is_contiguous(memory_format=torch.channels_last))
        except Exception as e:
            msg = 'miopen_convolution_relu failed: %r' % (str(e)[:120],)
        print('%s cin=%d cout=%d k=%d hw=%d: conv %.3f ms, conv+bias_act_ %.3f ms, %s' % (str(dtype)[6:], cin, cout, k, hw, t_conv, t_mine, msg), flush=True)
