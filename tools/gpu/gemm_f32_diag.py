"""Where the float32 1x1 GEMM's time goes: libraries built with -DOPA_GEMM_DIAG=1|2|3 drop the K loop's global loads, then its LDS
stores, then the epilogue's residual load and output store (wrong results, timing only).
    python tools/gpu/gemm_f32_diag.py      (builds the three libraries itself; each runs in a process of its own)"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from openpifpaf_amd import fused
B = 32
def t_ms(fn, reps=8):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
out = []
for name, hw, cin, cout, res in [('l1 conv3 64->256+res', 321, 64, 256, True), ('l1 conv1 256->64', 321, 256, 64, False),
                                 ('l2 conv3 128->512+res', 161, 128, 512, True), ('l2 conv1 512->128', 161, 512, 128, False),
                                 ('l3 conv3 256->1024+res', 81, 256, 1024, True), ('l3 conv1 1024->256', 81, 1024, 256, False),
                                 ('l4 conv3 512->2048+res', 41, 512, 2048, True), ('l4 conv1 2048->512', 41, 2048, 512, False)]:
    x = torch.randn((B, cin, hw, hw), device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, cin), device='cuda') * 0.05
    bias = torch.randn((cout,), device='cuda')
    r = torch.randn((B, cout, hw, hw), device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    t = t_ms(lambda: fused.conv1x1_bias_act(x, w, bias, r, True, None))
    out.append('%%.3f' %% t)
    del x, r; torch.cuda.empty_cache()
print(' '.join(out))
''' % (os.path.abspath(ROOT),)

from openpifpaf_amd import build  # noqa: E402
names = 'l1c3 l1c1 l2c3 l2c1 l3c3 l3c1 l4c3 l4c1'
ideal = []
for hw, cin, cout in ((321, 64, 256), (321, 256, 64), (161, 128, 512), (161, 512, 128), (81, 256, 1024), (81, 1024, 256), (41, 512, 2048), (41, 2048, 512)):
    ideal.append('%.3f' % (2.0 * 32 * hw * hw * cin * cout / 157.3e9))
print('%-44s %s' % ('ms per launch, batch 32:', names.replace(' ', '  ')))
print('%-44s %s' % ('MFMA alone at the peak', ' '.join(ideal)))
for diag, what in ((0, 'full'), (1, 'no global loads in the K loop'), (2, '... and no LDS stores'), (3, '... and no residual load / output store')):
    env = dict(os.environ)
    if diag:
        env['OPA_LIB_PATH'] = build.build_diagnostic('OPA_GEMM_DIAG=%d' % diag, 'gemmdiag%d' % diag, source='gemm_f32.hip', verbose=False)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print('%-44s %s' % (what, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED: ' + r.stderr[-300:]), flush=True)
