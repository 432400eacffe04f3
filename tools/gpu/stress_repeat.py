"""Repeat-decode stress: the same input decoded many times through one handle must give
bit-identical results that equal the oracle (catches cross-wave ordering races)."""
import sys
import numpy as np, torch
from openpifpaf_amd import native, synth, constants
from oracle import port
sk0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
nd = native.CifCaf(17, torch.from_numpy(sk0))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad_total = 0
for seed, people, size in ((32, 6, 65), (31, 3, 41), (33, 12, 81), (34, 20, 81), (35, 9, 65)):
    cif, caf = synth.synth_fields(seed, people, height=size, width=size)
    want, _ = port.decode(cif, 8, caf, 8, sk0)
    ct, ft = torch.from_numpy(cif).cuda()[None], torch.from_numpy(caf).cuda()[None]
    bad = 0
    for rep in range(reps):
        out, ids, counts = nd.call_batch(ct, 8, ft, 8)
        n = int(counts[0])
        if n != len(want) or np.abs(out[0, :n].cpu().numpy() - want).max() > 1e-4:
            bad += 1
    print('seed %d people %d size %d: %d poses, %d/%d mismatching runs' % (seed, people, size, len(want), bad, reps))
    bad_total += bad
# a full batch, repeated
cifs, cafs = synth.synth_batch(32, seed0=900, height=81, width=81)
ct, ft = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
first = None
for rep in range(50):
    out, ids, counts = nd.call_batch(ct, 8, ft, 8)
    cur = (out.cpu().clone(), counts.cpu().clone())
    if first is None:
        first = cur
    else:
        same = torch.equal(cur[1], first[1]) and all(
            torch.equal(cur[0][b, :int(cur[1][b])], first[0][b, :int(cur[1][b])]) for b in range(32))
        bad_total += 0 if same else 1
print('batch-32 repeat mismatches', bad_total)
sys.exit(1 if bad_total else 0)
