"""Phase timers of the association kernel (needs a library built with -DOPA_ASSOC_TIMING):
    OPA_LIB_PATH=/tmp/libopa_timing.so python tools/assoc_timing.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from openpifpaf_amd import constants, native, synth

B = 8
skel = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
cifs, cafs = synth.synth_batch(B, seed0=0)
dec = native.CifCaf(17, torch.from_numpy(skel))
cif, caf = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
for _ in range(3):
    out, ids, counts = dec.call_batch(cif, 8, caf, 8)
torch.cuda.synchronize()
scratch = dec.workspace_view('annotation_scratch', torch.int64).cpu().numpy()
per_image = dec.max_annotations * 17 * 4
names = ['blend', '#blend', '#blend_iters', 'seeds+rest', 'grow', 'mark', 'nms', 'total']
print('wall_clock64 ticks are 100 MHz (10 ns)')
for b in range(B):
    t = scratch[b * per_image:b * per_image + 12]
    n_seeds = int(dec.workspace_view('seed_count', torch.int32)[b])
    print('image %d people %2d poses %2d seeds %5d | ' % (b, synth.PEOPLE_CYCLE[b % 8], int(counts[b]), n_seeds)
          + '  '.join('%s=%d' % (n, v) for n, v in zip(names, t)) + '  shader clock %.0f MHz' % (t[8] / max(1, t[7]) * 100)
          + '  | us: total %.0f grow %.0f blend %.0f (%.2f us/blend, %.1f iters/blend) seeds %.0f mark %.0f nms %.0f rounds %d (stopped early %d) grown %d accepted %d select %.0f wait %.0f' % (
              t[7] / 100, t[4] / 100, t[0] / 100, t[0] / 100 / max(1, t[1]), t[2] / max(1, t[1]), t[3] / 100, t[5] / 100, t[6] / 100, t[9] & 0xffff, (t[9] >> 16) & 0xffff, (t[9] >> 32) & 0xffff, (t[9] >> 48) & 0xffff, t[10] / 100, t[11] / 100))
