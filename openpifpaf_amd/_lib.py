"""ctypes binding of ``libopenpifpaf_amd.so`` (C ABI: ``include/openpifpaf_amd.h``).

There is deliberately NO fallback: if the HIP library has not been built, or no
MI355X is visible, every compute entry point raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C openpifpaf_amd/csrc``.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OPA_LIB_PATH') or os.path.join(HERE, 'lib', 'libopenpifpaf_amd.so')

OPA_OK = 0
ERROR_NAMES = {1: 'INVALID_ARGUMENT', 2: 'HIP', 3: 'UNSUPPORTED', 4: 'WORKSPACE', 5: 'NO_DEVICE'}


class NativeLibraryMissing(ImportError):
    pass


class NativeError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """``opa_params``: the reference's process-global tunables
    (reference ``csrc/src/module.cpp:26-32,76-116``)."""
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

    def copy(self):
        other = Params()
        ctypes.memmove(ctypes.byref(other), ctypes.byref(self), ctypes.sizeof(Params))
        return other


ABI_VERSION = 6                      # OPA_ABI_VERSION of include/openpifpaf_amd.h


class Debug(ctypes.Structure):
    """``opa_debug``: A/B and test switches of one decoder handle (none changes a result).  The library reads the ``OPA_*``
    environment variables once, when it is loaded, into the defaults; a decode reads only its handle's copy."""
    _fields_ = [
        ('stage_worklist', ctypes.c_int32), ('fuse_scored', ctypes.c_int32), ('scored_one_pass', ctypes.c_int32),
        ('assoc_waves', ctypes.c_int32), ('assoc_growers', ctypes.c_int32), ('assoc_bbox', ctypes.c_int32),
        ('assoc_dedup', ctypes.c_int32), ('assoc_prededup', ctypes.c_int32), ('assoc_predict', ctypes.c_int32),
        ('assoc_predict_min_v', ctypes.c_float), ('assoc_predict_th', ctypes.c_float),
        ('assoc_collide', ctypes.c_int32), ('assoc_collide_shift', ctypes.c_int32), ('assoc_inherit', ctypes.c_int32),
        ('assoc_lookahead', ctypes.c_int32), ('assoc_help', ctypes.c_int32), ('assoc_spec', ctypes.c_int32),
        ('assoc_timing', ctypes.c_int32), ('assoc_persistent', ctypes.c_int32), ('fc_split', ctypes.c_int32),
        ('side_stream', ctypes.c_int32),
        ('assoc_watchdog_ticks', ctypes.c_int64),
    ]


class DetShape(ctypes.Structure):
    """``opa_det_shape``."""
    _fields_ = [(n, ctypes.c_int32) for n in ('batch', 'n_fields', 'field_h', 'field_w', 'stride', 'max_detections')]


class Shape(ctypes.Structure):
    """``opa_shape``."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        'batch', 'n_cif', 'n_caf', 'cif_h', 'cif_w', 'caf_h', 'caf_w',
        'cif_stride', 'caf_stride', 'max_annotations', 'n_keypoints', 'cifhr_pool_tiles')]


# every symbol include/openpifpaf_amd.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _dbl, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t
_P = ctypes.POINTER
SYMBOLS = {
    'opa_abi_version': (ctypes.c_int, []),
    'opa_shape_bytes': (_sz, []),
    'opa_params_bytes': (_sz, []),
    'opa_debug_bytes': (_sz, []),
    'opa_default_debug': (None, [_P(Debug)]),
    'opa_cifcaf_set_debug': (ctypes.c_int, [_vp, _P(Debug)]),
    'opa_cifcaf_get_debug': (ctypes.c_int, [_vp, _P(Debug)]),
    'opa_version': (ctypes.c_char_p, []),
    'opa_last_error': (ctypes.c_char_p, []),
    'opa_device_count': (ctypes.c_int, []),
    'opa_set_quiet': (None, [ctypes.c_int]),
    'opa_set_seed_tie_order': (None, [ctypes.c_int]),
    'opa_get_seed_tie_order': (ctypes.c_int, []),
    'opa_default_params': (None, [_P(Params)]),
    'opa_get_params': (None, [_P(Params)]),
    'opa_set_params': (ctypes.c_int, [_P(Params)]),
    'opa_cifcaf_create': (ctypes.c_int, [_P(_vp), _i32, _vp, _i32]),
    'opa_cifcaf_destroy': (None, [_vp]),
    'opa_cifcaf_get_state': (ctypes.c_int, [_vp, _P(_i32), _vp, _P(_i32)]),
    'opa_cifcaf_set_tie_placement': (ctypes.c_int, [_vp, _i32]),
    'opa_cifcaf_workspace_bytes': (_sz, [_P(Shape)]),
    'opa_cifcaf_workspace_bytes_for': (_sz, [_P(Shape), _P(Params)]),
    'opa_cifcaf_decode': (ctypes.c_int, [_vp, _P(Shape), _P(Params), _vp, _vp, _vp, _vp, _i32,
                                         _vp, _sz, _vp, _vp, _vp, _vp]),
    'opa_cifcaf_cifhr_view': (ctypes.c_int, [_P(Shape), _P(_sz), _P(_i32), _P(_i32), _P(_i32), _P(_dbl)]),
    'opa_cifcaf_get_cifhr': (ctypes.c_int, [_P(Shape), _vp, _i32, _vp, _vp]),
    'opa_cifcaf_workspace_view': (ctypes.c_int, [_P(Shape), ctypes.c_char_p, _P(_sz), _P(_sz)]),
    'opa_cifhr_pitch': (_i32, [_i32, _i32]),
    'opa_cifhr_scratch_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'opa_cifhr_accumulate': (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _dbl, _dbl, _P(Params),
                                            _vp, _vp, _sz, _vp]),
    'opa_cifseeds_scratch_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'opa_cifseeds_fill': (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _P(Params),
                                         _vp, _vp, _vp, _vp, _sz, _vp]),
    'opa_cifdetseeds_fill': (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _P(Params),
                                            _vp, _vp, _vp, _vp, _sz, _vp]),
    'opa_cafscored_fill': (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32,
                                          _vp, _dbl, _dbl, _P(Params), _vp, _vp, _vp]),
    'opa_grow_connection_blend': (ctypes.c_int, [_vp, _i32, _dbl, _dbl, _dbl, _dbl, _i32, _P(_dbl), _vp]),
    'opa_cifdet_workspace_bytes': (_sz, [_P(DetShape)]),
    'opa_cifdet_decode': (ctypes.c_int, [_P(DetShape), _P(Params), _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    'opa_bias_act': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    'opa_gemm_bias_act_bf16': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    'opa_gemm_pro_bias_act_bf16': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    'opa_gemm_bias_act_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    'opa_gemm_bias_act_f32x3': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    'opa_conv_rows_f32x3': (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 12 + [_vp]),
    'opa_conv3x3_f32x3': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'opa_gemm2_bias_act_f32x3': (ctypes.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'opa_conv3x3_winograd_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 8 + [_vp]),
    'opa_dwconv_bias_act': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'opa_channel_interleave': (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    'opa_head_epilogue': (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.c_uint32, _i32, _vp, _vp]),
    'opa_profile_begin': (ctypes.c_int, [_vp]),
    'opa_profile_end': (ctypes.c_int, [_i32, _P(ctypes.c_char_p), _P(ctypes.c_float), _P(_i32)]),
}

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Load the HIP library (once).  Raises NativeLibraryMissing if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                'openpifpaf_amd: %s not found. The HIP extension is required (there is no CPU path); '
                'build it with `python -c "import __graft_entry__ as g; g.build()"`.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(handle, name)     # AttributeError = ABI mismatch, fail loudly
            fn.restype = restype
            fn.argtypes = argtypes
        # the structs are passed by pointer: a library built from another header would read past (or short of) them
        if handle.opa_abi_version() != ABI_VERSION or handle.opa_shape_bytes() != ctypes.sizeof(Shape) or \
                handle.opa_params_bytes() != ctypes.sizeof(Params) or handle.opa_debug_bytes() != ctypes.sizeof(Debug):
            raise NativeError('openpifpaf_amd: %s was built from another include/openpifpaf_amd.h (ABI %d, opa_shape %d bytes, '
                              'opa_params %d bytes; this package: ABI %d, %d, %d): rebuild it' % (
                                  LIB_PATH, handle.opa_abi_version(), handle.opa_shape_bytes(), handle.opa_params_bytes(),
                                  ABI_VERSION, ctypes.sizeof(Shape), ctypes.sizeof(Params)))
        _lib = handle
    return _lib


def check(code, what=''):
    if code != OPA_OK:
        msg = lib().opa_last_error().decode('utf-8', 'replace')
        raise NativeError('%s failed: OPA_ERR_%s: %s' % (what or 'native call', ERROR_NAMES.get(code, code), msg))


def default_params(**overrides):
    p = Params()
    lib().opa_default_params(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError('opa_params has no field %r' % k)
        setattr(p, k, v)
    return p


def default_debug(**overrides):
    d = Debug()
    lib().opa_default_debug(ctypes.byref(d))
    for k, v in overrides.items():
        if not hasattr(d, k):
            raise AttributeError('opa_debug has no field %r' % k)
        setattr(d, k, v)
    return d


def get_params():
    p = Params()
    lib().opa_get_params(ctypes.byref(p))
    return p


def set_params(p):
    check(lib().opa_set_params(ctypes.byref(p)), 'opa_set_params')


def profile_begin(stream_ptr):
    check(lib().opa_profile_begin(stream_ptr), 'opa_profile_begin')


def profile_end(capacity=64):
    """-> list of (kernel name, milliseconds) in launch order."""
    names = (ctypes.c_char_p * capacity)()
    ms = (ctypes.c_float * capacity)()
    n = ctypes.c_int32()
    check(lib().opa_profile_end(capacity, names, ms, ctypes.byref(n)), 'opa_profile_end')
    return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, capacity))]
