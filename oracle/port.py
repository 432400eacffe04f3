"""ctypes binding of the plain-C++ oracle (``oracle/cifcaf_oracle.cpp``).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libcifcaf_oracle.so')


class Params(ctypes.Structure):
    """Mirror of ``Params`` in cifcaf_oracle.cpp (= the reference's static members)."""
    _fields_ = [
        ('cif_threshold', ctypes.c_double),
        ('cifhr_neighbors', ctypes.c_int64),
        ('seed_threshold', ctypes.c_double),
        ('caf_threshold', ctypes.c_double),
        ('cif_floor', ctypes.c_double),
        ('keypoint_threshold', ctypes.c_double),
        ('keypoint_threshold_rel', ctypes.c_double),
        ('nms_suppression', ctypes.c_double),
        ('nms_instance_threshold', ctypes.c_double),
        ('nms_keypoint_threshold', ctypes.c_double),
        ('force_complete_caf_th', ctypes.c_double),
        ('occupancy_reduction', ctypes.c_double),
        ('occupancy_min_scale', ctypes.c_double),
        ('greedy', ctypes.c_int32),
        ('reverse_match', ctypes.c_int32),
        ('force_complete', ctypes.c_int32),
        ('block_joints', ctypes.c_int32),
        ('ablation_cifseeds_nms', ctypes.c_int32),
        ('ablation_cifseeds_no_rescore', ctypes.c_int32),
        ('ablation_caf_no_rescore', ctypes.c_int32),
        ('ablation_cifhr_skip', ctypes.c_int32),
    ]


_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_cifseeds.restype = ctypes.c_int64
        _lib.oracle_cifdetseeds.restype = ctypes.c_int64
        _lib.oracle_cifcaf_decode.restype = ctypes.c_int64
        _lib.oracle_cifcaf_decode_k.restype = ctypes.c_int64
        _lib.oracle_cifdet_decode.restype = ctypes.c_int64
    return _lib


def sorted_seed_order(scores):
    """perm[k] = original position of the seed std::sort (cif_seeds.cpp:94) leaves at rank k."""
    v = _f32(scores).ravel()
    perm = np.zeros(len(v), dtype=np.int64)
    lib().oracle_sorted_seed_order(_ptr(v), _i64(len(v)), _ptr(perm))
    return perm


def set_seed_tie_rule(rule):
    """Order of seeds with exactly equal scores in the restatement: 0 = libstdc++'s unstable std::sort
    (the reference, default), 1 = cell index ascending (the HIP path's total order), 2 = descending."""
    lib().oracle_set_seed_tie_rule(ctypes.c_int(int(rule)))


def default_params(**overrides):
    p = Params()
    lib().oracle_default_params(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(v):
    return ctypes.c_int64(int(v))


def cifhr_accumulate(cif, stride, *, min_scale=0.0, factor=1.0, params=None):
    """-> raw CifHr buffer [F,Hhr,Whr] at revision 1 (0 untouched, else 1+value)."""
    params = params or default_params()
    cif = _f32(cif)
    F, C, H, W = cif.shape
    assert C == 5
    hr = np.zeros((F, (H - 1) * stride + 1, (W - 1) * stride + 1), dtype=np.float32)
    lib().oracle_cifhr_accumulate(_ptr(cif), _i64(F), _i64(H), _i64(W), _i64(stride),
                                  ctypes.c_double(min_scale), ctypes.c_double(factor),
                                  ctypes.byref(params), _ptr(hr))
    return hr


def cifseeds(cif, stride, cifhr, *, params=None):
    """-> (f int64[n], vxys float32[n,4]) sorted by v descending (std::sort)."""
    params = params or default_params()
    cif, cifhr = _f32(cif), _f32(cifhr)
    F, _, H, W = cif.shape
    cap = F * H * W
    out_f = np.empty((cap,), dtype=np.int64)
    out_v = np.empty((cap, 4), dtype=np.float32)
    n = lib().oracle_cifseeds(_ptr(cif), _i64(F), _i64(H), _i64(W), _i64(stride), _ptr(cifhr),
                              ctypes.byref(params), _ptr(out_f), _ptr(out_v), _i64(cap))
    return out_f[:n].copy(), out_v[:n].copy()


def cifdetseeds(field, stride, cifhr, *, params=None):
    """CifDetSeeds (cif_seeds.cpp:69-90,117-139) -> (f int64[n], vxywh float32[n,5]) sorted by v descending."""
    params = params or default_params()
    field, cifhr = _f32(field), _f32(cifhr)
    F, C, H, W = field.shape
    assert C == 6
    cap = F * H * W
    out_f = np.empty((cap,), dtype=np.int64)
    out_v = np.empty((cap, 5), dtype=np.float32)
    n = lib().oracle_cifdetseeds(_ptr(field), _i64(F), _i64(H), _i64(W), _i64(stride), _ptr(cifhr),
                                 ctypes.byref(params), _ptr(out_f), _ptr(out_v), _i64(cap))
    return out_f[:n].copy(), out_v[:n].copy()


def cafscored(caf, stride, cifhr, cif_shape, cif_stride, skeleton0, *,
              score_th=-1.0, cif_floor=0.1, params=None):
    """-> (forward, backward): lists (one per CAF field) of float32 [n,7] arrays."""
    params = params or default_params()
    caf, cifhr = _f32(caf), _f32(cifhr)
    A, C, H, W = caf.shape
    assert C == 8
    F, _, cH, cW = cif_shape
    skel = np.ascontiguousarray(skeleton0, dtype=np.int64)
    cap = H * W
    fwd = np.empty((A, cap, 7), dtype=np.float32)
    bwd = np.empty((A, cap, 7), dtype=np.float32)
    nf = np.zeros((A,), dtype=np.int32)
    nb = np.zeros((A,), dtype=np.int32)
    lib().oracle_cafscored(_ptr(caf), _i64(A), _i64(H), _i64(W), _i64(stride), _ptr(cifhr),
                           _i64(F), _i64(cH), _i64(cW), _i64(cif_stride), _ptr(skel),
                           ctypes.c_double(score_th), ctypes.c_double(cif_floor),
                           ctypes.byref(params), _i64(cap), _ptr(fwd), _ptr(nf), _ptr(bwd), _ptr(nb))
    return ([fwd[a, :nf[a]].copy() for a in range(A)], [bwd[a, :nb[a]].copy() for a in range(A)])


def grow_connection_blend(rows, x, y, s, filter_sigmas=1.0, only_max=False):
    """-> (x, y, s, v) like ``torch.ops.openpifpaf_decoder.grow_connection_blend``."""
    rows = _f32(rows).reshape(-1, 7)
    out = np.zeros((4,), dtype=np.float64)
    lib().oracle_grow_connection_blend(_ptr(rows), _i64(rows.shape[0]), ctypes.c_double(x),
                                       ctypes.c_double(y), ctypes.c_double(s),
                                       ctypes.c_double(filter_sigmas), ctypes.c_int32(int(only_max)),
                                       _ptr(out))
    return out


def decode(cif, cif_stride, caf, caf_stride, skeleton0, *, params=None,
           initial_annotations=None, initial_ids=None, return_cifhr=False, cap=4096, n_keypoints=None):
    """The whole CifCaf decode -> (annotations float32 [n,K,4] (v,x,y,s), ids int64[n]).
    ``n_keypoints`` > number of CIF fields: the reference's tracking setup (tracking_pose.py:47-80)."""
    params = params or default_params()
    cif, caf = _f32(cif), _f32(caf)
    F, _, H, W = cif.shape
    A, _, cH, cW = caf.shape
    skel = np.ascontiguousarray(skeleton0, dtype=np.int64)
    assert skel.shape == (A, 2)
    K = int(n_keypoints) if n_keypoints else F
    assert K >= F
    out = np.zeros((cap, K, 4), dtype=np.float32)
    ids = np.full((cap,), -1, dtype=np.int64)
    hr = None
    if return_cifhr:
        hr = np.zeros((F, (H - 1) * cif_stride + 1, (W - 1) * cif_stride + 1), dtype=np.float32)
    if initial_annotations is not None and len(initial_annotations):
        init = _f32(initial_annotations)
        init_ids = np.ascontiguousarray(initial_ids, dtype=np.int64)
        n_init = init.shape[0]
        init_p, ids_p = _ptr(init), _ptr(init_ids)
    else:
        n_init, init_p, ids_p = 0, None, None
    n = lib().oracle_cifcaf_decode_k(
        _ptr(cif), _i64(F), _i64(H), _i64(W), _i64(cif_stride),
        _ptr(caf), _i64(A), _i64(cH), _i64(cW), _i64(caf_stride),
        _ptr(skel), ctypes.byref(params), init_p, ids_p, _i64(n_init),
        _i64(cap), _ptr(out), _ptr(ids), _ptr(hr) if hr is not None else None, _i64(K))
    if n > cap:
        raise RuntimeError('oracle produced %d annotations > cap %d' % (n, cap))
    res = (out[:n].copy(), ids[:n].copy())
    if return_cifhr:
        res = res + (hr,)
    return res


def cifdet_decode(field, stride, *, params=None, max_detections=120, return_cifhr=False):
    """CifDet::call (cifdet.cpp:24-80) -> (categories int64[n], scores float32[n], boxes float32[n,4])."""
    params = params or default_params()
    field = _f32(field)
    F, C, H, W = field.shape
    assert C == 6
    cat = np.zeros((max_detections,), dtype=np.int64)
    sc = np.zeros((max_detections,), dtype=np.float32)
    bx = np.zeros((max_detections, 4), dtype=np.float32)
    hr = np.zeros((F, (H - 1) * stride + 1, (W - 1) * stride + 1), dtype=np.float32) if return_cifhr else None
    n = lib().oracle_cifdet_decode(_ptr(field), _i64(F), _i64(H), _i64(W), _i64(stride), ctypes.byref(params),
                                   _i64(max_detections), _ptr(cat), _ptr(sc), _ptr(bx),
                                   _ptr(hr) if hr is not None else None)
    res = (cat[:n].copy(), sc[:n].copy(), bx[:n].copy())
    return res + (hr,) if return_cifhr else res
