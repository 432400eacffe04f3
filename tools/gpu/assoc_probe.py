"""Statistics of the association kernel on the bench batch (always-on counters, "assoc_stats"):
    [OPA_ASSOC_WAVES=8|12|16] python tools/gpu/assoc_probe.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from openpifpaf_amd import _lib, constants, native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
skel = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
cifs, cafs = synth.synth_batch(B, seed0=0)
dec = native.CifCaf(17, torch.from_numpy(skel))
cif, caf = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
for _ in range(3):
    out, ids, counts = dec.call_batch(cif, 8, caf, 8)
torch.cuda.synchronize()
st = dec.assoc_stats().cpu().numpy()
print('OPA_ASSOC_WAVES=%s  ticks are 10 ns' % os.environ.get('OPA_ASSOC_WAVES', 'default'))
print('img people poses seeds | started accepted stopped dropped pre-stop mispred refills | growth-us total-us iters '
      'head-wait-us grower-busy-us scans us/scan growers')
for b in range(min(B, 8)):
    s = st[b]
    print('%3d %6d %5d %5d | %7d %8d %9d %7d %8d %7d %7d | %9.0f %8.0f %6.0f %13.0f %14.0f %5d %7.2f %7d' % (
        b, synth.PEOPLE_CYCLE[b % 8], native.count_rows(int(counts[b])), s[7], s[0], s[1], s[2], s[3], s[4], s[5], s[6],
        s[8] / 100, s[9] / 100, s[15], s[12] / 100, s[10] / 100, s[11], s[10] / 100 / max(1, s[11]), s[13]))
tot = st.sum(axis=0)
print('coordinator us (image: wait-iters wait commit refill hand-out other): ' + '  '.join('%d: %d %.0f %.0f %.0f %.0f %.0f' % (b, st[b][16], st[b][12] / 100, st[b][17] / 100, st[b][18] / 100, st[b][19] / 100, (st[b][8] - st[b][12] - st[b][17] - st[b][18] - st[b][19]) / 100) for b in range(min(B, 8))))
print('grower time per scan (image: all-in us, inside the scan us, of that until loads returned us): ' + '  '.join('%d: %.2f %.2f %.2f' % (b, st[b][10] / 100 / max(1, st[b][11]), st[b][21] / 100 / max(1, st[b][11]), st[b][22] / 100 / max(1, st[b][11])) for b in range(min(B, 8))))
print('refill us spent waiting for the growers\' occupancy marks (image: us): ' + '  '.join('%d: %.0f' % (b, st[b][20] / 100) for b in range(min(B, 8))))
print('batch: started %d accepted %d cancelled %d dropped %d -> discarded share %.1f%% of growths; '
      'slowest image %.0f us, mean %.0f us' % (tot[0], tot[1], tot[2], tot[3],
                                               100.0 * (tot[2] + tot[3] + tot[4]) / max(1, tot[0]),
                                               st[:, 9].max() / 100, st[:, 9].mean() / 100))
acc = {}
for _ in range(10):
    _lib.profile_begin(native._stream())
    dec.call_batch(cif, 8, caf, 8)
    for name, ms in _lib.profile_end():
        acc.setdefault(name, []).append(ms)
print('  '.join('%s %.1f us' % (k.replace('_kernel', ''), 1e3 * np.mean(v)) for k, v in acc.items()),
      ' | decode %.3f ms' % sum(np.mean(v) for v in acc.values()))
if os.environ.get('OPA_TRACE_IMAGE'):
    b = int(os.environ['OPA_TRACE_IMAGE'])
    tr = dec.workspace_view('assoc_trace', torch.int32).view(B, 64, 4)[b].cpu().numpy()
    n = int(st[b][1])
    print('image %d: commit#  seed  grower  handed-out-us  done-us  commit-us  (growth us, waited-for-growth us)' % b)
    prev = 0
    for k in range(min(n, 64)):
        tc, te, td, sg = [int(v) for v in tr[k]]
        print('  %3d %6d %3d %9.1f %9.1f %9.1f   growth %6.1f  head waited %6.1f' % (
            k, sg & 0xFFFFFF, sg >> 24, te / 100, td / 100, tc / 100, (td - te) / 100, max(0, td - prev) / 100))
        prev = tc
