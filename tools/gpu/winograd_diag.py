"""Where a chunk's time goes in the eight-wave Winograd kernel: a library built with -DOPA_WINO_DIAG runs it without the filter
reloads (11), without the pixel fetches (12), without the transforms (13) -- wrong results, timing only.
    python tools/gpu/winograd_diag.py      (builds lib/libopenpifpaf_amd_winodiag.so itself)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from openpifpaf_amd import build  # noqa: E402

os.environ['OPA_LIB_PATH'] = build.build_diagnostic('OPA_WINO_DIAG=1', 'winodiag', source='winograd.hip')
import torch  # noqa: E402
from openpifpaf_amd import winograd  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from winograd_probe import time_ms  # noqa: E402

winograd.VARIANTS.update({v: (16, 2) for v in range(11, 19)})
B = 32
for (C, H) in ((64, 321), (128, 161), (256, 81), (512, 41)):
    x = torch.randn((B, C, H, H), device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
    u = winograd.transform_filter(w, 2)
    out = torch.empty_like(x)
    line = 'C %3d %3dx%3d:' % (C, H, H)
    for variant, name in ((2, 'full'), (11, 'no filter reloads'), (12, 'no pixel fetches'), (13, 'no transforms'),
                          (14, 'MFMA + operand reads only'), (15, 'MFMA only'), (16, 'no output transform'),
                          (17, 'output transform without its stores'), (18, 'nontemporal stores')):
        t = time_ms(lambda: winograd.conv3x3(x, u, C, variant=variant, out=out), 10)
        line += '  %s %.3f ms' % (name, t)
    ideal = 2.0 * B * H * H * C * C * 9 / 2.25 / 157.3e9
    print(line + '  (MFMA alone at the peak: %.3f ms)' % ideal, flush=True)
