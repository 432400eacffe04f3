"""Where the force-complete kernel's time goes (library built with -DOPA_FC_TIMING): per image the longest growth phase of a
workgroup, the longest single pose and the keypoint NMS of the last workgroup."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpifpaf_amd import _lib, constants, native, synth
skel0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
params = _lib.default_params(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0, nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
cifs, cafs = synth.synth_batch(32, seed0=0)
cd, fd = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
dec = native.CifCaf(17, torch.from_numpy(skel0))
for i in range(3):
    if i == 2:
        dec.workspace_view('assoc_trace', torch.int32).zero_()
    out, ids, counts = dec.call_batch(cd, 8, fd, 8, params=params)
torch.cuda.synchronize()
h = dec.workspace_view('assoc_trace', torch.int32)[40 * 4:40 * 4 + 35].cpu().numpy()
print('scans by chunks hit (0..31, 32+):', h[:33].tolist())
print('scans: %d, of them compacted %d with %.1f entries on average' % (h[:33].sum(), h[33], h[34] / max(h[33], 1)))
tr_all = dec.workspace_view('assoc_trace', torch.int32)[:32 * 64 * 4].view(32, 64, 4).cpu().numpy()
tr = tr_all[:, 56]
longest = tr_all[:, 57, 0]
print('longest pose per image (us, scans):', [(round((int(v) >> 10) / 100.0, 1), int(v) & 1023) for v in longest[:12]])
order = np.argsort(-tr[:, 0])
print('img poses | growth-phase us  longest-pose us  nms us')
for b in order[:10]:
    print('%3d %5d | %8.1f %8.1f %8.1f' % (b, tr[b, 3], tr[b, 0] / 100.0, tr[b, 1] / 100.0, tr[b, 2] / 100.0))
