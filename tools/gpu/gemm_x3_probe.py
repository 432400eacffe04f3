"""The float32 1x1 convolution three ways -- the float32 MFMA kernel (csrc/gemm_f32.hip), the split-operand kernel on the bf16
MFMA pipe with nine and with six partial products (csrc/gemm_f32x3.hip) -- and torch's own float32 convolution: time per launch
and the error against a float64 product on the ResNet-50 shapes at 641 px / batch 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from openpifpaf_amd import fused  # noqa: E402

SHAPES = [  # (name, H = W, K, N, residual)
    ('layer1 reduce', 321, 256, 64, False), ('layer1 expand', 321, 64, 256, True),
    ('layer2 reduce', 161, 512, 128, False), ('layer2 expand', 161, 128, 512, True),
    ('layer3 reduce', 81, 1024, 256, False), ('layer3 expand', 81, 256, 1024, True),
    ('layer4 reduce', 41, 2048, 512, False), ('layer4 expand', 41, 512, 2048, True),
    ('head', 41, 2048, 320, False),
]
B = int(os.environ.get('B', '32'))
if os.environ.get('SHAPES'):
    SHAPES = [sh for sh in SHAPES if any(sh[0].startswith(f) for f in os.environ['SHAPES'].split(','))]


def time_ms(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


torch.manual_seed(0)
for name, hw, K, N, res in SHAPES:
    if hw == 321 and B > 32:
        continue
    x = torch.randn((B, K, hw, hw), device='cuda').clamp_(min=0).contiguous(memory_format=torch.channels_last)   # (post-ReLU activations)
    w = torch.randn((N, K), device='cuda') * (2.0 / K) ** 0.5
    bias = torch.randn(N, device='cuda') * 0.1
    r = torch.randn((B, N, hw, hw), device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    w3 = fused.split_weight(w)
    assert torch.equal(w3.float().sum(0), w), 'split_weight is not exact'
    M = B * hw * hw
    flops = 2.0 * M * K * N
    # float64 reference on a sample of rows
    rows = torch.randint(0, M, (4096,), device='cuda')
    xa = x.permute(0, 2, 3, 1).reshape(M, K)
    ref = xa[rows].double() @ w.double().t() + bias.double()
    if res:
        ref = ref + r.permute(0, 2, 3, 1).reshape(M, N)[rows].double()
    ref = ref.clamp_(min=0)
    scale = float(ref.abs().max())
    line = '%-14s M %8d K %4d N %4d:' % (name, M, K, N)
    outs = {}
    for label, fn in (
            ('f32 mfma', lambda: fused.conv1x1_bias_act(x, w, bias, r, True)),
            ('x3/9', lambda: fused.conv1x1_bias_act_x3(x, w3, bias, r, True, terms=9)),
            ('x3/6', lambda: fused.conv1x1_bias_act_x3(x, w3, bias, r, True, terms=6)),
            ('torch', lambda: torch.relu_(torch.nn.functional.conv2d(x, w.view(N, K, 1, 1), bias) + (r if res else 0)))):
        ms = time_ms(fn)
        out = fn().permute(0, 2, 3, 1).reshape(M, N)[rows].double()
        d = (out - ref).abs()
        outs[label] = out
        line += '  %s %.3f ms %5.1f TF err max %.2e rms %.2e |' % (label, ms, flops / ms * 1e-9, float(d.max()) / scale,
                                                                   float((d * d).mean().sqrt()) / scale)
    print(line, flush=True)
