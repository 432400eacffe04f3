"""Generates tests/golden/cifcaf_golden.npz from the REAL reference decoder.

Run in the build container (needs /root/reference -> oracle/_ref/openpifpaf_ref.so):
    python tests/golden/make_golden.py

The reference's own test-suite holds no known-answer vectors for the decode path
(SURVEY.md 8c), so the golden vectors are outputs of the reference itself (its
unmodified C++ compiled by oracle/build_ref.py, a FRESH decoder instance per case)
on seeded synthetic fields.  Inputs are not stored (6 MB per image); they are
regenerated from the seed by openpifpaf_amd.synth and pinned by a SHA-256 of their
bytes, so a change of the generator or of numpy's RNG stream is detected instead
of silently invalidating the fixtures.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from common import DET_CASES, GOLDEN_CASES          # noqa: E402
from openpifpaf_amd import constants, synth   # noqa: E402
from oracle import port, reference       # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def as_u8(text):
    return np.frombuffer(text.encode(), dtype=np.uint8)


def main():
    torch = reference.load()
    torch.set_num_threads(1)
    reference.reset_statics()
    skel = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
    out = {}
    fc = port.default_params(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                             nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
    U = torch.classes.openpifpaf_decoder_utils
    for i, (seed, people, H, W) in enumerate(GOLDEN_CASES):
        cif, caf = synth.synth_fields(seed, people, height=H, width=W)
        reference.reset_statics()
        ann, ids, hr = reference.decode(cif, 8, caf, 8, skel)
        out['case%d_input_sha256' % i] = as_u8(digest(cif, caf))
        out['case%d_annotations' % i] = ann
        out['case%d_ids' % i] = ids
        out['case%d_cifhr_sha256' % i] = as_u8(digest(hr))
        # stage-level goldens from the reference's utility classes
        cif_t, caf_t = torch.from_numpy(cif), torch.from_numpy(caf)
        chr_ = U.CifHr()
        chr_.reset(list(cif.shape), 8)
        chr_.accumulate(cif_t, 8, 0.0, 1.0)
        acc, rev = chr_.get_accumulated()
        seeds = U.CifSeeds(acc, rev)
        seeds.fill(cif_t, 8)
        sf, sv = seeds.get()
        out['case%d_seed_f' % i] = sf.numpy().copy()
        out['case%d_seed_vxys' % i] = sv.numpy().copy()
        cs = U.CafScored(acc, rev, -1.0, 0.1)
        cs.fill(caf_t, 8, torch.from_numpy(skel))
        fwd, bwd = cs.get()
        out['case%d_caf_counts' % i] = np.array([[len(f), len(b)] for f, b in zip(fwd, bwd)], dtype=np.int32)
        out['case%d_caf_sha256' % i] = as_u8(
            digest(*[f.numpy() for f in fwd], *[b.numpy() for b in bwd]))
        # force-complete variant (the reference's benchmark setting)
        reference.apply_params(fc)
        ann_fc, _, _ = reference.decode(cif, 8, caf, 8, skel)
        reference.reset_statics()
        out['case%d_annotations_fc' % i] = ann_fc
        print('case %d seed=%d people=%d %dx%d: %d poses (%d force-complete), %d seeds'
              % (i, seed, people, H, W, len(ann), len(ann_fc), len(sf)))
    # CifDet (reference torch.classes.openpifpaf_decoder.CifDet, fresh instance per case)
    for i, (seed, n_obj, H, W) in enumerate(DET_CASES):
        field = synth.synth_det_field(seed, n_obj, height=H, width=W)
        det = torch.classes.openpifpaf_decoder.CifDet()
        c, sc, bx = det.call(torch.from_numpy(field), 8)
        out['det%d_input_sha256' % i] = as_u8(digest(field))
        out['det%d_categories' % i] = c.numpy().copy()
        out['det%d_scores' % i] = sc.numpy().copy()
        out['det%d_boxes' % i] = bx.numpy().copy()
        print('det case %d seed=%d objects=%d %dx%d: %d detections' % (i, seed, n_obj, H, W, len(c)))
    # grow_connection_blend known answers
    cif, caf = synth.synth_fields(7, 10)
    hr = port.cifhr_accumulate(cif, 8)
    fwd, _ = port.cafscored(caf, 8, hr, cif.shape, 8, skel)
    rows = max(fwd, key=len)
    rng = np.random.default_rng(5)
    queries, answers = [], []
    for _ in range(32):
        r = rows[rng.integers(len(rows))]
        q = (float(r[1] + rng.normal(0, 1)), float(r[2] + rng.normal(0, 1)), float(rng.uniform(2, 30)),
             float(rng.choice([1.0, 4.0])), bool(rng.integers(2)))
        queries.append(q)
        answers.append(torch.ops.openpifpaf_decoder.grow_connection_blend(torch.from_numpy(rows), *q))
    out['blend_rows'] = rows
    out['blend_queries'] = np.asarray(queries, dtype=np.float64)
    out['blend_answers'] = np.asarray(answers, dtype=np.float64)
    path = os.path.join(HERE, 'cifcaf_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
