"""Random shapes through the split-operand kernels (1x1, block tail + downsampling, 3x3 with padding 1 and stride 1-3, stem, head)
against float64: every output element, batches that end inside a 128-row tile, odd image sizes, channel counts at the kernels'
limits.  python tools/gpu/x3_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from openpifpaf_amd import fused  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.manual_seed(rng.randrange(1 << 30))
worst = {}


def check(kind, got, ref, desc):
    d = (got.double() - ref).abs()
    rel = float(d.max()) / max(float(ref.abs().max()), 1e-30)
    worst[kind] = max(worst.get(kind, 0.0), rel)
    if not (rel < 3e-6) or tuple(got.shape) != tuple(ref.shape):
        print('MISMATCH', kind, desc, rel, tuple(got.shape), tuple(ref.shape), flush=True)
        return False
    return True


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


bad = 0
with torch.no_grad():
    for case in range(n_cases):
        kind = rng.choice(['gemm', 'pair', 'conv3', 'stem', 'head'])
        B = rng.choice([1, 2, 3, 5])
        for terms in (6, 9):
            fused.X3_TERMS = terms
            if kind == 'gemm':
                K, N = 64 * rng.randint(1, 12), 64 * rng.randint(1, 9)
                h, w = rng.randint(1, 40), rng.randint(1, 40)
                x = cl(torch.randn(B, K, h, w, device='cuda'))
                wt = torch.randn(N, K, device='cuda') / K ** 0.5
                bias = torch.randn(N, device='cuda')
                res = cl(torch.randn(B, N, h, w, device='cuda')) if rng.random() < 0.5 else None
                ab = torch.randn(K, device='cuda') if rng.random() < 0.4 else None
                relu = rng.random() < 0.7
                got = fused.conv1x1_bias_act_x3(x, fused.split_weight(wt), bias, res, relu, ab, terms)
                xd = x.double() if ab is None else (x.double() + ab.double().view(1, -1, 1, 1)).clamp_(min=0)
                ref = torch.nn.functional.conv2d(xd, wt.double().view(N, K, 1, 1), bias.double())
                if res is not None:
                    ref = ref + res.double()
                ref = ref.clamp_(min=0) if relu else ref
                ok = check(kind, got, ref, (B, K, N, h, w, res is not None, ab is not None, relu, terms))
            elif kind == 'pair':
                K1 = 32 * rng.randint(1, 10)
                K2 = 32 * rng.randint(1, 10)
                if (K1 + K2) % 64:
                    K2 += 32
                N, s = 64 * rng.randint(1, 8), rng.choice([1, 1, 2, 2, 3])
                hi, wi = rng.randint(1, 45), rng.randint(1, 45)
                ho, wo = (hi - 1) // s + 1, (wi - 1) // s + 1
                x = cl(torch.randn(B, K2, hi, wi, device='cuda').clamp_(min=0))
                hh = cl(torch.randn(B, K1, ho, wo, device='cuda'))
                conv = torch.nn.Conv2d(K1, N, 1, bias=False).cuda()
                dconv = torch.nn.Conv2d(K2, N, 1, s, bias=False).cuda()
                bias = torch.randn(N, device='cuda')
                ab = torch.randn(K1, device='cuda') if rng.random() < 0.5 else None
                assert fused.pair_supported(conv, dconv, hh, x, bias, ab)
                got = fused.conv1x1_pair_bias_act_x3(conv, dconv, hh, x, bias, True, ab)
                hd = hh.double() if ab is None else (hh.double() + ab.double().view(1, -1, 1, 1)).clamp_(min=0)
                ref = (torch.nn.functional.conv2d(hd, conv.weight.double()) + torch.nn.functional.conv2d(x.double(), dconv.weight.double(), stride=s)
                       + bias.double().view(1, -1, 1, 1)).clamp_(min=0)
                ok = check(kind, got, ref, (B, K1, K2, N, hi, wi, s, ab is not None, terms))
            elif kind == 'conv3':
                C, N, s = 64 * rng.randint(1, 6), 64 * rng.randint(1, 6), rng.choice([1, 2, 2, 3])
                hi, wi = rng.randint(1, 40), rng.randint(1, 40)
                x = cl(torch.randn(B, C, hi, wi, device='cuda'))
                conv = torch.nn.Conv2d(C, N, 3, s, 1, bias=False).cuda()
                bias = torch.randn(N, device='cuda')
                relu = rng.random() < 0.7
                assert fused.conv3x3_x3_supported(conv, x, bias)
                got = fused.conv3x3_bias_act_x3(conv, x, bias, relu)
                ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), stride=s, padding=1) + bias.double().view(1, -1, 1, 1)
                ref = ref.clamp_(min=0) if relu else ref
                ok = check(kind, got, ref, (B, C, N, hi, wi, s, relu, terms))
            elif kind == 'stem':
                N = 64 * rng.randint(1, 3)
                hi, wi = rng.randint(1, 120), rng.randint(1, 120)
                x = torch.randn(B, 3, hi, wi, device='cuda')
                if rng.random() < 0.5:
                    x = cl(x)
                conv = torch.nn.Conv2d(3, N, 7, 2, 3, bias=False).cuda()
                bias = torch.randn(N, device='cuda')
                assert fused.stem_x3_supported(conv, x, bias)
                got = fused.stem7x7_bias_act_x3(conv, x, bias, True)
                ref = (torch.nn.functional.conv2d(x.double(), conv.weight.double(), stride=2, padding=3) + bias.double().view(1, -1, 1, 1)).clamp_(min=0)
                ok = check(kind, got, ref, (B, N, hi, wi, terms))
            else:
                K, N = 64 * rng.randint(1, 10), rng.randint(1, 700)
                h, w = rng.randint(1, 30), rng.randint(1, 30)
                x = cl(torch.randn(B, K, h, w, device='cuda'))
                conv = torch.nn.Conv2d(K, N, 1, bias=rng.random() < 0.8).cuda()
                assert fused.head_conv_x3_supported(conv, x)
                got = fused.head_conv_x3(conv, x)
                ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), None if conv.bias is None else conv.bias.double())
                ok = check(kind, got, ref, (B, K, N, h, w, terms))
            bad += 0 if ok else 1
torch.cuda.synchronize()
print('x3 fuzz: %d cases x {6, 9} terms, %d mismatches; worst max error / max |ref| per kind: %s'
      % (n_cases, bad, {k: float('%.2g' % v) for k, v in sorted(worst.items())}))
sys.exit(1 if bad else 0)
