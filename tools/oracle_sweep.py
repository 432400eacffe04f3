"""Randomised sweep of the oracle restatement (oracle/cifcaf_oracle.cpp through oracle/port.py) against the REAL
reference decoder (oracle/_ref, built from the reference's own C++ sources): the same case generator as
tools/gpu/parity_sweep.py, every output compared bit for bit.  Runs on the CPU, where /root/reference exists.

    PYTHONPATH=. python tools/oracle_sweep.py [n_images] [seed] [coco|dense|tracking|wholebody|cifdet]

Exit code 1 on the first difference.  The log of the committed run is profiles/r1/oracle_vs_reference_sweep.log."""
import sys
import time

import numpy as np

from openpifpaf_amd import constants, synth
from oracle import port, reference

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mode = sys.argv[3] if len(sys.argv) > 3 else 'coco'
if not reference.available():
    sys.exit('oracle/_ref is not built (needs /root/reference): python -c "import __graft_entry__ as g; g.build()"')
reference.load().set_num_threads(1)
if mode == 'cifdet':                     # CifDet::call (cifdet.cpp:24-80), fresh instance per image
    torch = reference.load()
    n_det = 0
    t0 = time.time()
    try:
        for image_i in range(n_images):
            H, W = int(rng.integers(12, 91)), int(rng.integers(12, 91))     # synth_det_field needs >= 9 cells
            stride = int(rng.choice([4, 8, 8, 16]))
            kw = [dict(), dict(), dict(cif_threshold=0.2), dict(seed_threshold=0.35), dict(cifhr_neighbors=9)][
                int(rng.integers(5))]
            seed_i = int(rng.integers(1 << 30))
            field = synth.synth_det_field(seed_i, int(rng.integers(0, 13)), n_categories=int(rng.integers(1, 12)),
                                          height=H, width=W)
            p = port.default_params(**kw)
            reference.apply_params(p)
            torch.classes.openpifpaf_decoder_utils.CifDetSeeds.set_threshold(p.seed_threshold)
            c, sc, bx = torch.classes.openpifpaf_decoder.CifDet().call(torch.from_numpy(field), stride)
            oc, osc, obx = port.cifdet_decode(field, stride, params=p)
            if not (np.array_equal(c.numpy(), oc) and np.array_equal(sc.numpy(), osc) and np.array_equal(bx.numpy(), obx)):
                print('DIFFERENCE image %d: %dx%d stride %d seed %d options %s' % (image_i, H, W, stride, seed_i, kw))
                sys.exit(1)
            n_det += len(oc)
    finally:
        reference.reset_statics()
    print('oracle == reference (cifdet): %d images, %d detections, bit-equal categories, scores and boxes, %.1f s' % (
        n_images, n_det, time.time() - t0))
    sys.exit(0)
wb = constants.wholebody() if mode == 'wholebody' else None
skeleton1 = {'wholebody': wb['skeleton'] if wb else None, 'coco': list(constants.COCO_PERSON_SKELETON),
             'dense': list(constants.COCO_PERSON_SKELETON) + list(constants.DENSER_COCO_PERSON_CONNECTIONS),
             'tracking': synth.tracking_skeleton()}[mode]
skel0 = np.asarray(skeleton1, dtype=np.int64) - 1
K = {'tracking': 34, 'wholebody': 133}.get(mode, 17)
# only what the reference exposes as statics (its occupancy reduction is a constant)
OPTIONS = [dict(), dict(), dict(greedy=1), dict(reverse_match=0), dict(block_joints=1),
           dict(keypoint_threshold=0.3, keypoint_threshold_rel=0.7),
           dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0, nms_instance_threshold=0.0,
                nms_keypoint_threshold=0.0),
           dict(cif_threshold=0.2, seed_threshold=0.25, caf_threshold=0.2), dict(cifhr_neighbors=9),
           dict(ablation_cifseeds_nms=1), dict(ablation_cifseeds_no_rescore=1), dict(ablation_caf_no_rescore=1),
           dict(ablation_cifhr_skip=1), dict(nms_suppression=0.5)]
n_poses = 0
t0 = time.time()
try:
    for image_i in range(n_images):
        H, W = int(rng.integers(7, 91)), int(rng.integers(7, 91))
        stride = int(rng.choice([4, 8, 8, 8, 16]))
        kw = OPTIONS[int(rng.integers(len(OPTIONS)))]
        lo = float(rng.uniform(0.15, 0.7))
        people = int(rng.integers(0, 1 + max(1, min(24, H * W // 120))))
        seed_i = int(rng.integers(1 << 30))
        init = init_ids = None
        if mode == 'tracking':
            cif, caf, full = synth.synth_tracking_fields(seed_i, people, height=H, width=W,
                                                         size_range=(lo, min(1.0, lo + 0.4)))
            n_init = int(rng.integers(0, 4))
            if n_init:
                prev, _ = port.decode(full, stride, caf, stride, skel0)
                init = np.zeros((n_init, 34, 4), dtype=np.float32)
                init[:min(n_init, len(prev)), 17:] = prev[:n_init, 17:]
                init_ids = np.arange(50, 50 + n_init, dtype=np.int64)
        else:
            cif, caf = synth.synth_fields(seed_i, min(people, 6) if wb else people, height=H, width=W,
                                          skeleton=skeleton1, pose=wb['standing_pose'] if wb else None,
                                          noise=float(rng.uniform(0.0, 0.4)), size_range=(lo, min(1.3, lo + 0.6)))
        p = port.default_params(**kw)
        reference.apply_params(p)
        r_out, r_ids, r_hr = reference.decode(cif, stride, caf, stride, skel0, n_keypoints=K,
                                              initial_annotations=init, initial_ids=init_ids)
        o_out, o_ids, o_hr = port.decode(cif, stride, caf, stride, skel0, params=p, n_keypoints=K,
                                         initial_annotations=init, initial_ids=init_ids, return_cifhr=True)
        same = (r_out.shape == o_out.shape and np.array_equal(r_out, o_out) and np.array_equal(r_ids, o_ids)
                and np.array_equal(r_hr, o_hr))
        if not same:
            print('DIFFERENCE image %d: %dx%d stride %d people %d seed %d options %s: reference %s vs oracle %s' % (
                image_i, H, W, stride, people, seed_i, kw, r_out.shape, o_out.shape))
            sys.exit(1)
        n_poses += len(r_out)
finally:
    reference.reset_statics()
print('oracle == reference (%s): %d images, %d poses, bit-equal annotations, ids and CifHr maps, %.1f s' % (
    mode, n_images, n_poses, time.time() - t0))
