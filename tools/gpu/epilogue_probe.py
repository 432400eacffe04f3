"""Streaming rate of the fused bias/residual/ReLU epilogue on ResNet-50's float32 / bfloat16 activations."""
import sys, time
import torch
sys.path.insert(0, '.')
from openpifpaf_amd import fused
for dt in (torch.float32, torch.bfloat16):
    for (c, hw) in [(256, 161), (512, 81), (1024, 41), (64, 161)]:
        x = torch.randn(32, c, hw, hw, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
        r = torch.randn_like(x)
        b = torch.randn(c, device='cuda').to(dt)
        for res in (r, None):
            for _ in range(2): fused.bias_act_(x, b, res, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fused.bias_act_(x, b, res, True)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
            nbytes = x.numel() * x.element_size() * (3 if res is not None else 2)
            print('%s C %4d hw %3d residual %d: %.3f ms, %.2f TB/s' % (dt, c, hw, res is not None, ms, nbytes / ms / 1e9))
