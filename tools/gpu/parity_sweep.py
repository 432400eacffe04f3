"""Randomised parity sweep of the HIP decode against the oracle: shapes, crowd sizes, strides and decoder
options drawn at random, several images per launch.  Exit code 1 on the first mismatch.

    PYTHONPATH=. python tools/gpu/parity_sweep.py [n_batches] [seed] [coco|dense|tracking|wholebody]

coco: 17 joints / 19 bones (register-resident growth state); dense: + 25 dense bones (LDS growth state);
tracking: 34 joints over 17 CIF fields, 36 bones, previous-frame poses as initial annotations;
wholebody: 133 joints / 160 bones."""
import sys
import time

import numpy as np
import torch

from openpifpaf_amd import _lib, constants, native, synth
from oracle import port, reference

# round 6: against the REAL reference (oracle/_ref, the reference's own csrc compiled by oracle/build_ref.py: it travels to the GPU
# box with the snapshot) wherever the option set can be pushed into its statics; the restatement (pinned bit-equal to it on the
# CPU, tests/test_oracle_vs_reference.py) for the options that are constructor constants there
USE_REF = reference.available()
REF_FIELDS = {'greedy', 'reverse_match', 'keypoint_threshold', 'keypoint_threshold_rel', 'force_complete', 'force_complete_caf_th',
              'cif_threshold', 'seed_threshold', 'caf_threshold', 'ablation_cifseeds_no_rescore', 'ablation_caf_no_rescore',
              'ablation_cifseeds_nms', 'ablation_cifhr_skip', 'nms_suppression', 'nms_instance_threshold', 'nms_keypoint_threshold',
              'cifhr_neighbors', 'block_joints'}
if USE_REF:
    reference.load().set_num_threads(1)
    reference.reset_statics()
n_ref = n_port = 0

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mode = sys.argv[3] if len(sys.argv) > 3 else 'coco'
wb = constants.wholebody() if mode == 'wholebody' else None
skeleton1 = {'wholebody': wb['skeleton'] if wb else None, 'coco': list(constants.COCO_PERSON_SKELETON),
             'dense': list(constants.COCO_PERSON_SKELETON) + list(constants.DENSER_COCO_PERSON_CONNECTIONS),
             'tracking': synth.tracking_skeleton()}[mode]
skel0 = np.asarray(skeleton1, dtype=np.int64) - 1
K = {'tracking': 34, 'wholebody': 133}.get(mode, 17)
dec = native.CifCaf(K, torch.from_numpy(skel0), max_annotations=512)
OPTIONS = [dict(), dict(), dict(greedy=1), dict(reverse_match=0), dict(keypoint_threshold=0.3, keypoint_threshold_rel=0.7),
           dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0, nms_instance_threshold=0.0,
                nms_keypoint_threshold=0.0),
           dict(cif_threshold=0.2, seed_threshold=0.25, caf_threshold=0.2), dict(ablation_cifseeds_no_rescore=1),
           dict(ablation_caf_no_rescore=1), dict(occupancy_reduction=1.0), dict(nms_suppression=0.5)]
n_images = n_poses = n_overflows = 0
worst = 0.0
t0 = time.time()
for batch_i in range(n_batches):
    H, W = int(rng.integers(7, 91)), int(rng.integers(7, 91))
    B = int(rng.integers(1, 7))
    stride = int(rng.choice([4, 8, 8, 8, 16]))
    kw = OPTIONS[int(rng.integers(len(OPTIONS)))]
    lo = float(rng.uniform(0.15, 0.7))
    cifs, cafs, inits = [], [], []
    n_init = int(rng.integers(0, 4)) if mode == 'tracking' else 0
    for b in range(B):
        people = int(rng.integers(0, 1 + max(1, min(24, H * W // 120))))
        seed_b = int(rng.integers(1 << 30))
        if mode == 'tracking':
            cif, caf, full = synth.synth_tracking_fields(seed_b, people, height=H, width=W,
                                                         size_range=(lo, min(1.0, lo + 0.4)))
            prev, _ = port.decode(full, stride, caf, stride, skel0)          # previous-frame poses: joints 17..33
            init = np.zeros((n_init, 34, 4), dtype=np.float32)
            init[:min(n_init, len(prev)), 17:] = prev[:n_init, 17:]
            inits.append(init)
        else:
            cif, caf = synth.synth_fields(seed_b, min(people, 6) if wb else people, height=H, width=W,
                                          skeleton=skeleton1, pose=wb['standing_pose'] if wb else None,
                                          noise=float(rng.uniform(0.0, 0.4)), size_range=(lo, min(1.3, lo + 0.6)))
        cifs.append(cif); cafs.append(caf)
    init_ids = np.tile(np.arange(50, 50 + n_init, dtype=np.int64), (B, 1))
    out, ids, cnt = dec.call_batch(
        torch.from_numpy(np.stack(cifs)).cuda(), stride, torch.from_numpy(np.stack(cafs)).cuda(), stride,
        torch.from_numpy(np.stack(inits)).cuda() if n_init else None,
        torch.from_numpy(init_ids).cuda() if n_init else None, params=_lib.default_params(**kw))
    out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
    if (cnt & native.COUNT_FAILED).any() and dec.pool_overflowed():
        # an image's CIF map reached more tiles than the automatic pool holds (big people on a small field): flagged, status
        # -2 -- what the synchronous entry points do on their own: once more with a pool that holds the whole map
        n_overflows += 1
        dec.use_full_pool()
        out, ids, cnt = dec.call_batch(
            torch.from_numpy(np.stack(cifs)).cuda(), stride, torch.from_numpy(np.stack(cafs)).cuda(), stride,
            torch.from_numpy(np.stack(inits)).cuda() if n_init else None,
            torch.from_numpy(init_ids).cuda() if n_init else None, params=_lib.default_params(**kw))
        out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
        dec.cifhr_pool_tiles = 0                      # (back to the automatic pool for the next shapes)
    native.check_counts(cnt)
    by_ref = USE_REF and set(kw) <= REF_FIELDS
    if by_ref:
        reference.apply_params(port.default_params(**kw))
    for b in range(B):
        if by_ref:
            want = reference.decode(cifs[b], stride, cafs[b], stride, skel0, n_keypoints=K,
                                    initial_annotations=inits[b] if n_init else None,
                                    initial_ids=init_ids[b] if n_init else None)[0]
            n_ref += 1
        else:
            want, _ = port.decode(cifs[b], stride, cafs[b], stride, skel0, params=port.default_params(**kw),
                                  n_keypoints=K, initial_annotations=inits[b] if n_init else None,
                                  initial_ids=init_ids[b] if n_init else None)
            n_port += 1
        n = int(cnt[b])
        if n & native.COUNT_OVERFLOW:          # capacity overflow is flagged, not compared
            assert len(want) > native.count_rows(n)
            continue
        ok = n == len(want)
        err = float(np.abs(out[b, :n].astype(np.float64) - want).max()) if ok and n else 0.0
        presence = ok and np.array_equal(out[b, :n, :, 0] > 0, want[:, :, 0] > 0)
        if not ok or err > 1e-4 or not presence:
            print('MISMATCH batch %d image %d: %dx%d stride %d options %s: %d poses vs oracle %d, max err %g' % (
                batch_i, b, H, W, stride, kw, n, len(want), err))
            sys.exit(1)
        worst = max(worst, err)
        n_images += 1
        n_poses += n
    if by_ref:
        reference.reset_statics()
print('parity sweep (' + mode + ') ok: %d launches, %d images, %d poses, worst |delta| %.3g, %.1f s; %d launches repeated with a full '
      'tile pool after a flagged overflow; %d images against the reference itself (oracle/_ref), %d against the restatement' % (
          n_batches, n_images, n_poses, worst, time.time() - t0, n_overflows, n_ref, n_port))
