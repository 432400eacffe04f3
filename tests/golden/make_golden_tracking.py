"""Generates tests/golden/tracking_golden.npz from the REAL reference (oracle/_ref, fresh decoder per case).

The reference's tracking setup (decoder/tracking_pose.py:47-80,163-217): a CifCaf object with 2*17 keypoints
and the single-frame skeleton + 17 temporal bones decodes a 17-field CIF of the current frame and the
concatenated CAF heads, starting from the previous frame's poses as initial annotations.

    python tests/golden/make_golden_tracking.py       (build container only: needs /root/reference)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from common import TRACKING_CASES, tracking_problem        # noqa: E402
from oracle import reference                               # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.hexdigest().encode(), dtype=np.uint8)


def main():
    torch = reference.load()
    torch.set_num_threads(1)
    reference.reset_statics()
    out = {}
    for i, case in enumerate(TRACKING_CASES):
        cif, caf, skel0, init, ids = tracking_problem(*case)
        ann, got_ids, _ = reference.decode(cif, 8, caf, 8, skel0, n_keypoints=34,
                                           initial_annotations=init, initial_ids=ids)
        ann0, _, _ = reference.decode(cif, 8, caf, 8, skel0, n_keypoints=34)
        out['case%d_input_sha256' % i] = digest(cif, caf, init)
        out['case%d_annotations' % i] = ann
        out['case%d_ids' % i] = got_ids
        out['case%d_annotations_no_initial' % i] = ann0
        print('tracking case %d %s: %d initial poses -> %d tracking poses (ids %s), %d without initial annotations'
              % (i, case, len(init), len(ann), got_ids.tolist(), len(ann0)))
    path = os.path.join(HERE, 'tracking_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
