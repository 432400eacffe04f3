"""Winograd F(2x2, 3x3) kernel (csrc/winograd.hip) against torch's convolution: error against a float64 convolution beside
the error of torch's own float32 result, then time per launch for the ResNet-50 shapes at 641 px / batch 32.
    python tools/gpu/winograd_probe.py [--quick] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from openpifpaf_amd import winograd  # noqa: E402


def time_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def check(B, C, O, H, W, variant, order, seed=0, bias=False, relu=False):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn((B, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn((O, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn((O,), generator=g).cuda() if bias else None
    u = winograd.transform_filter(w, variant)
    y = winograd.conv3x3(x, u, O, bias=b, relu=relu, variant=variant, order=order)
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    if relu:
        ref64 = ref64.relu()
    ref32 = torch.nn.functional.conv2d(x, w, b, padding=1)
    if relu:
        ref32 = ref32.relu()
    scale = ref64.abs().max().item()
    e_w = (y.double() - ref64).abs().max().item() / scale
    e_t = (ref32.double() - ref64).abs().max().item() / scale
    ok = e_w < 2e-5
    print('check B%d C%d O%d %dx%d variant %d order %d bias %d relu %d: winograd %.2e  torch-f32 %.2e (max |err| / max |y|) %s'
          % (B, C, O, H, W, variant, order, bias, relu, e_w, e_t, 'ok' if ok else 'MISMATCH'), flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--variants', default='0,1,2')
    ap.add_argument('--timing-only', action='store_true')
    ap.add_argument('--orders', default='0,1')
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(',')]
    ok = True
    for variant in ([] if args.timing_only else variants):
        for (B, C, O, H, W) in ((1, 16, 64, 8, 8), (2, 32, 64, 7, 9), (3, 64, 128, 21, 21), (2, 64, 64, 41, 40), (1, 128, 128, 5, 3)):
            for order in (0, 1):
                ok &= check(B, C, O, H, W, variant, order)
        ok &= check(2, 32, 64, 11, 13, variant, 0, bias=True, relu=True)
    if not ok:
        print('MISMATCH'); sys.exit(1)
    if args.quick:
        return
    torch.backends.cudnn.benchmark = True
    B = args.batch
    for (C, H) in ((64, 321), (128, 161), (256, 81), (512, 41)):
        x = torch.randn((B, C, H, H), device='cuda').contiguous(memory_format=torch.channels_last)
        w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).cuda().to(memory_format=torch.channels_last)
        conv.weight.data.copy_(w)
        flops = 2.0 * B * H * H * C * C * 9
        with torch.no_grad():
            t_ref = time_ms(lambda: conv(x), args.reps)
            line = 'C %3d %3dx%3d: torch/MIOpen %.3f ms (%.1f TFLOP/s)' % (C, H, H, t_ref, flops / t_ref / 1e9)
            yref = conv(x)
            for variant in variants:
                u = winograd.transform_filter(w, variant)
                for order in [int(o) for o in args.orders.split(',')]:
                    out = torch.empty_like(yref)
                    t = time_ms(lambda: winograd.conv3x3(x, u, C, variant=variant, order=order, out=out), args.reps)
                    err = (out - yref).abs().max().item() / yref.abs().max().item()
                    line += ' | v%d o%d %.3f ms (%.1f direct-equivalent TFLOP/s, MFMA %.1f; |d| %.1e)' % (
                        variant, order, t, flops / t / 1e9, flops / 2.25 / t / 1e9, err)
        print(line, flush=True)


if __name__ == '__main__':
    main()
