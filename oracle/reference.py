"""Loader and thin helpers for the REAL reference decoder, ``_ref/openpifpaf_ref.so``.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The shared object is the reference's unmodified ``csrc/src/*.cpp`` compiled by
``oracle/build_ref.py``.  It registers the reference's TorchScript classes:
``torch.classes.openpifpaf_decoder.CifCaf`` and
``torch.classes.openpifpaf_decoder_utils.{CifHr,CifSeeds,CafScored,Occupancy,...}``
(reference ``csrc/src/module.cpp:19-118``).

Hygiene: a NEW ``CifCaf`` per image for parity (revision drift),
``set_quiet(True)``, statics reset to the reference defaults by :func:`reset_statics`.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, '_ref', 'openpifpaf_ref.so')

_loaded = False


def available():
    return os.path.exists(SO_PATH)


def load():
    global _loaded
    import torch
    if not _loaded:
        if not available():
            raise RuntimeError('oracle/_ref/openpifpaf_ref.so missing: run python oracle/build_ref.py '
                               '(needs /root/reference)')
        torch.ops.load_library(SO_PATH)
        torch.ops.openpifpaf.set_quiet(True)
        _loaded = True
    return torch


def reset_statics():
    """Reference defaults: cifcaf.cpp:18-24, cif_hr.cpp:13-15, cif_seeds.cpp:11-14,
    caf_scored.cpp:11-12, nms_keypoints.cpp:12-14."""
    torch = load()
    C = torch.classes.openpifpaf_decoder.CifCaf
    C.set_block_joints(False); C.set_greedy(False)
    C.set_keypoint_threshold(0.15); C.set_keypoint_threshold_rel(0.5)
    C.set_reverse_match(True); C.set_force_complete(False); C.set_force_complete_caf_th(0.001)
    U = torch.classes.openpifpaf_decoder_utils
    U.CifHr.set_neighbors(16); U.CifHr.set_threshold(0.3); U.CifHr.set_ablation_skip(False)
    U.CifSeeds.set_threshold(0.2); U.CifSeeds.set_ablation_nms(False); U.CifSeeds.set_ablation_no_rescore(False)
    U.CifDetSeeds.set_threshold(0.2)
    U.CafScored.set_default_score_th(0.3); U.CafScored.set_ablation_no_rescore(False)
    U.NMSKeypoints.set_instance_threshold(0.15); U.NMSKeypoints.set_keypoint_threshold(0.15)
    U.NMSKeypoints.set_suppression(0.00001)


def apply_params(p):
    """Push an ``oracle.port.Params`` into the reference's process-global statics."""
    torch = load()
    C = torch.classes.openpifpaf_decoder.CifCaf
    C.set_block_joints(bool(p.block_joints)); C.set_greedy(bool(p.greedy))
    C.set_keypoint_threshold(p.keypoint_threshold); C.set_keypoint_threshold_rel(p.keypoint_threshold_rel)
    C.set_reverse_match(bool(p.reverse_match)); C.set_force_complete(bool(p.force_complete))
    C.set_force_complete_caf_th(p.force_complete_caf_th)
    U = torch.classes.openpifpaf_decoder_utils
    U.CifHr.set_neighbors(p.cifhr_neighbors); U.CifHr.set_threshold(p.cif_threshold)
    U.CifHr.set_ablation_skip(bool(p.ablation_cifhr_skip))
    U.CifSeeds.set_threshold(p.seed_threshold); U.CifSeeds.set_ablation_nms(bool(p.ablation_cifseeds_nms))
    U.CifSeeds.set_ablation_no_rescore(bool(p.ablation_cifseeds_no_rescore))
    U.CafScored.set_default_score_th(p.caf_threshold)
    U.CafScored.set_ablation_no_rescore(bool(p.ablation_caf_no_rescore))
    U.NMSKeypoints.set_instance_threshold(p.nms_instance_threshold)
    U.NMSKeypoints.set_keypoint_threshold(p.nms_keypoint_threshold)
    U.NMSKeypoints.set_suppression(p.nms_suppression)


def new_decoder(n_keypoints, skeleton0):
    torch = load()
    return torch.classes.openpifpaf_decoder.CifCaf(int(n_keypoints), torch.as_tensor(skeleton0, dtype=torch.int64))


def decode(cif, cif_stride, caf, caf_stride, skeleton0, *, initial_annotations=None, initial_ids=None,
           n_keypoints=None):
    """Fresh decoder instance -> (annotations [n,K,4] np.float32, ids np.int64, cifhr np.float32 raw)."""
    torch = load()
    import numpy as np
    dec = new_decoder(n_keypoints or cif.shape[0], skeleton0)
    cif_t = torch.from_numpy(np.ascontiguousarray(cif, dtype=np.float32))
    caf_t = torch.from_numpy(np.ascontiguousarray(caf, dtype=np.float32))
    if initial_annotations is not None and len(initial_annotations):
        ia = torch.from_numpy(np.ascontiguousarray(initial_annotations, dtype=np.float32))
        ii = torch.from_numpy(np.ascontiguousarray(initial_ids, dtype=np.int64))
    else:
        ia, ii = None, None
    out, ids = dec.call_with_initial_annotations(cif_t, int(cif_stride), caf_t, int(caf_stride), ia, ii)
    hr, rev = dec.get_cifhr()
    assert rev == 1.0
    return out.numpy().copy(), ids.numpy().copy(), hr.numpy().copy()
