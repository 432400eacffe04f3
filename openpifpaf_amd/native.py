"""Torch-facing wrappers over the C ABI, mirroring the reference's TorchScript
classes (reference ``csrc/src/module.cpp:19-118``):

===============================================  =============================================
reference                                        here
===============================================  =============================================
``torch.ops.openpifpaf.set_quiet``               :func:`set_quiet`
``torch.classes.openpifpaf_decoder.CifCaf``      :class:`CifCaf` (+ ``call_batch``)
``torch.ops.openpifpaf_decoder.grow_connection_blend``  :func:`grow_connection_blend`
``torch.classes.openpifpaf_decoder_utils.CifHr``        :class:`CifHr`
``torch.classes.openpifpaf_decoder_utils.CifSeeds``     :class:`CifSeeds`
``torch.classes.openpifpaf_decoder_utils.CafScored``    :class:`CafScored`
``...NMSKeypoints`` / static get_/set_ pairs     static get_/set_ pairs on the same classes
===============================================  =============================================

PyTorch is used for device memory and streams only; all compute is in the HIP
library.  Tensors given on the CPU are uploaded to the current device first (the
reference only accepts CPU tensors, ``cifcaf.cpp:137-138``); results come back
on the device the inputs were on.
"""
import ctypes
import os

import torch

from . import _lib

DEFAULT_MAX_ANNOTATIONS = 128
COUNT_OVERFLOW = 0x40000000          # OPA_COUNT_OVERFLOW (include/openpifpaf_amd.h)
COUNT_FAILED = 0x20000000            # OPA_COUNT_FAILED: the association kernel's watchdog gave up on the image
COUNT_ROWS_MASK = 0x0FFFFFFF         # OPA_COUNT_ROWS


def count_rows(counts):
    """``counts`` as returned by ``call_batch`` (tensor, numpy array or int) -> number of valid rows."""
    return counts & COUNT_ROWS_MASK


def count_overflowed(counts):
    """-> truthy where poses were dropped for lack of annotation capacity."""
    return (counts & COUNT_OVERFLOW) != 0


def count_failed(counts):
    """-> truthy where the association kernel gave up on the image (``OPA_COUNT_FAILED``)."""
    return (counts & COUNT_FAILED) != 0


def check_counts(counts):
    """``counts`` ON THE HOST (numpy array, CPU tensor or int; they travel with the result's one D2H copy): raise
    when the association kernel's watchdog fired for an image -- such an image reports no poses, and handing that
    on as "nobody in the picture" would be a silent wrong answer."""
    import numpy as np
    c = np.asarray(counts.cpu() if hasattr(counts, 'cpu') else counts).reshape(-1)
    bad = np.nonzero(c & COUNT_FAILED)[0]
    if len(bad):
        raise _lib.NativeError('openpifpaf_amd: the decode of image(s) %s of the batch failed (status -1: the association '
                               'kernel\'s watchdog fired; status -2: the image\'s CIF map reaches more tiles than its pool holds -- '
                               'CifCaf(..., cifhr_pool_tiles=\'full\') or use_full_pool()); their result is invalid' % bad.tolist())


def set_quiet(quiet=True):
    _lib.lib().opa_set_quiet(int(bool(quiet)))


SEED_TIE_ORDERS = {'libstdcxx': 1, 'index': 0, 'libstdcxx-fused': 2}


def set_seed_tie_order(order='libstdcxx'):
    """Order of seeds with EQUAL scores: ``'libstdcxx'`` (default) = what the reference's unstable ``std::sort``
    leaves (cif_seeds.cpp:94), reproduced on the device for the images that have such seeds; ``'index'`` = cell index
    ascending (no such pass); ``'libstdcxx-fused'`` = the same as ``'libstdcxx'`` since round 6 (WHERE the pass runs -- inside
    the association kernel by default, or as a launch of its own -- is a decoder's choice: :meth:`CifCaf.set_tie_placement`).
    Process-global, like the reference's statics."""
    _lib.lib().opa_set_seed_tie_order(SEED_TIE_ORDERS[order])


def get_seed_tie_order():
    return {0: 'index', 1: 'libstdcxx', 2: 'libstdcxx-fused'}[_lib.lib().opa_get_seed_tie_order()]


def _device():
    if not torch.cuda.is_available():
        raise _lib.NativeError('openpifpaf_amd: no MI355X/HIP device visible; the decode path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _prep(t, dtype=torch.float32):
    """-> (contiguous device tensor, original device)"""
    orig = t.device
    if t.device.type != 'cuda':
        t = t.to(_device(), non_blocking=True)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous(), orig


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _static_getset(field, cast=float):
    def getter():
        return cast(getattr(_lib.get_params(), field))

    def setter(v):
        p = _lib.get_params()
        setattr(p, field, cast(v) if cast is not bool else int(bool(v)))
        _lib.set_params(p)
    return staticmethod(getter), staticmethod(setter)


class CifCaf:
    """Drop-in for ``torch.classes.openpifpaf_decoder.CifCaf`` (module.cpp:25-54)."""

    get_block_joints, set_block_joints = _static_getset('block_joints', bool)
    get_greedy, set_greedy = _static_getset('greedy', bool)
    get_keypoint_threshold, set_keypoint_threshold = _static_getset('keypoint_threshold')
    get_keypoint_threshold_rel, set_keypoint_threshold_rel = _static_getset('keypoint_threshold_rel')
    get_reverse_match, set_reverse_match = _static_getset('reverse_match', bool)
    get_force_complete, set_force_complete = _static_getset('force_complete', bool)
    get_force_complete_caf_th, set_force_complete_caf_th = _static_getset('force_complete_caf_th')

    def __init__(self, n_keypoints, skeleton, *, max_annotations=DEFAULT_MAX_ANNOTATIONS, cifhr_pool_tiles=0):
        """``cifhr_pool_tiles``: capacity per image of the high-resolution map, which the decode keeps as a pool of 32x64
        tiles (``opa_shape::cifhr_pool_tiles``): 0 = automatic (the whole map where that is at most 32 MB per image; else an
        eighth of it, at least 1024 tiles, plus one spill region for the batch that holds the rest of ONE whole map: 8 MB for a
        641-px COCO image instead of 31), ``'full'`` / -1 = every tile, n > 0 = n tiles.  An image whose CIF cells reach
        more tiles than the pool holds is flagged (``OPA_COUNT_FAILED``, status -2) instead of decoded wrongly; the
        synchronous entry points (``call``, ``call_with_initial_annotations``, ``decoder.CifCaf.batch``) then decode once
        more with a full pool on their own, ``call_batch`` (asynchronous) leaves that to the caller (:meth:`use_full_pool`)."""
        skeleton = torch.as_tensor(skeleton)
        if skeleton.dtype != torch.int64:
            raise RuntimeError('skeleton must be of type LongTensor')      # cifcaf.hpp:106
        self.n_keypoints = int(n_keypoints)
        self.skeleton = skeleton.detach().cpu().contiguous().reshape(-1, 2)
        self.max_annotations = int(max_annotations)
        self.cifhr_pool_tiles = -1 if cifhr_pool_tiles in ('full', -1) else int(cifhr_pool_tiles)
        self._handle = ctypes.c_void_p()
        self._workspaces = {}
        self._last = None
        self._pinned = False                 # a captured graph holds raw pointers into the workspace
        self._workspace_fc = False           # the live workspace has the force-complete regions
        _device()
        _lib.check(_lib.lib().opa_cifcaf_create(
            ctypes.byref(self._handle), self.n_keypoints,
            ctypes.c_void_p(self.skeleton.data_ptr()), self.skeleton.shape[0]), 'opa_cifcaf_create')

    def __del__(self):
        try:
            if getattr(self, '_handle', None):
                _lib.lib().opa_cifcaf_destroy(self._handle)
                self._handle = None
        except Exception:   # interpreter shutdown
            pass

    # pickle state = (n_keypoints, skeleton), module.cpp:41-53
    def set_tie_placement(self, inside_association):
        """Where the pass that puts seeds of equal score into the reference's order runs for this decoder: ``True`` inside the
        association kernel (every image its own ties), ``False`` a launch of its own (its time then shows up under its own
        name), ``None`` (default) automatic = inside the kernel: measured shorter for one decode at a time and for several in
        flight (round 6).  Same results bit for bit (``opa_cifcaf_set_tie_placement``)."""
        _lib.check(_lib.lib().opa_cifcaf_set_tie_placement(
            self._handle, -1 if inside_association is None else int(bool(inside_association))), 'opa_cifcaf_set_tie_placement')

    def set_debug(self, **switches):
        """A/B and test switches of THIS decoder (``opa_debug``, ``include/openpifpaf_amd.h``): exact variants of the kernels,
        the watchdog, measurements -- none changes a result.  ``set_debug(assoc_growers=3)``; ``set_debug()`` with no
        argument restores the defaults (the library's, with the ``OPA_*`` environment variables it read when it was loaded).
        Nothing reads the environment during a decode."""
        if switches:
            d = self.get_debug()
            for k, v in switches.items():
                if not hasattr(d, k):
                    raise AttributeError('opa_debug has no field %r' % k)
                setattr(d, k, v)
        else:
            d = _lib.default_debug()
        _lib.check(_lib.lib().opa_cifcaf_set_debug(self._handle, ctypes.byref(d)), 'opa_cifcaf_set_debug')

    def get_debug(self):
        d = _lib.Debug()
        _lib.check(_lib.lib().opa_cifcaf_get_debug(self._handle, ctypes.byref(d)), 'opa_cifcaf_get_debug')
        return d

    def __getstate__(self):
        return (self.n_keypoints, self.skeleton, self.max_annotations, self.cifhr_pool_tiles)

    def __setstate__(self, state):
        self.__init__(state[0], state[1], max_annotations=state[2], cifhr_pool_tiles=state[3] if len(state) > 3 else 0)

    def _shape(self, cif, cif_stride, caf, caf_stride):
        B, F, C, H, W = cif.shape
        Bc, A, Cc, cH, cW = caf.shape
        if C != 5 or Cc != 8 or B != Bc:
            raise ValueError('expected cif [B,F,5,H,W] and caf [B,A,8,H,W], got %s and %s'
                             % (tuple(cif.shape), tuple(caf.shape)))
        if F > self.n_keypoints:
            raise ValueError('the CIF field has %d fields, more than the decoder\'s %d keypoints' % (F, self.n_keypoints))
        # F < n_keypoints: tracking setup, joints F.. have no CIF field (reference tracking_pose.py:47-80)
        return _lib.Shape(B, F, A, H, W, cH, cW, int(cif_stride), int(caf_stride), self.max_annotations,
                          self.n_keypoints, self.cifhr_pool_tiles)

    def _workspace(self, shape, device, params=None):
        """The decoder's workspace for this shape; the regions only a force-complete decode uses (a second set of CAF
        lists: ~30 % of it) are allocated only once such a decode is asked for."""
        fc = bool((params if params is not None else _lib.get_params()).force_complete)
        key = tuple(getattr(shape, n) for n, _ in _lib.Shape._fields_) + (device.index,)
        ws = self._workspaces.get(key)
        if ws is not None and fc and not self._workspace_fc:
            if self._pinned:
                raise _lib.NativeError('this decoder has a captured HIP graph that replays into a workspace without the '
                                       'force-complete regions: use another CifCaf instance')
            ws = None                        # grow: the lazy tile clear starts over with the new block
        if ws is None:
            fc_params = _lib.default_params(force_complete=int(fc))
            nbytes = _lib.lib().opa_cifcaf_workspace_bytes_for(ctypes.byref(shape), ctypes.byref(fc_params))
            if nbytes == 0:
                raise _lib.NativeError(_lib.lib().opa_last_error().decode())
            if self._pinned:
                raise _lib.NativeError('this decoder has a captured HIP graph that replays into its workspace: '
                                       'use another CifCaf instance for a different shape')
            self._workspaces.clear()        # one live workspace per decoder
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            # The block comes from the caching allocator and may hold anything, including a stale header
            # of an earlier workspace: the lazy tile clear trusts the header, so it starts out invalid
            # (include/openpifpaf_amd.h, workspace contract).
            ws[:256].zero_()
            self._workspaces[key] = ws
            self._workspace_fc = fc
        return ws

    def use_full_pool(self):
        """From the next decode on the map's tile pool holds every tile (after an image did not fit the automatic one)."""
        if self._pinned:
            raise _lib.NativeError('this decoder has a captured HIP graph that replays into its workspace: use another CifCaf instance')
        self.cifhr_pool_tiles = -1

    def pool_overflowed(self):
        """Did an image of the last ``call_batch`` reach more map tiles than the pool holds?  (Synchronises.)"""
        if self._last is None:
            return False
        shape, _ = self._last
        if shape.cifhr_pool_tiles == -1:         # (the pool of THAT call; the setting may have changed since)
            return False
        return bool(self.workspace_view('cifhr_overflow', torch.int32)[:shape.batch].any().item())

    def call_batch(self, cif, cif_stride, caf, caf_stride, initial_annotations=None, initial_ids=None,
                   *, params=None):
        """Batched, asynchronous decode on the current stream.

        :param cif: ``[B,F,5,H,W]`` float32, :param caf: ``[B,A,8,H,W]`` float32 (device tensors)
        :returns: ``(annotations [B,max,K,4] (v,x,y,s), ids [B,max] int64, counts [B] int32)``
                  device tensors.  ``count_rows(counts[b])`` rows are valid, the rest is undefined;
                  ``count_overflowed(counts[b])``: poses were dropped because ``max_annotations`` is too small.
        """
        cif, orig = _prep(cif)
        caf, _ = _prep(caf)
        shape = self._shape(cif, cif_stride, caf, caf_stride)
        ws = self._workspace(shape, cif.device, params)
        B, K = shape.batch, self.n_keypoints
        out = torch.empty((B, self.max_annotations, K, 4), dtype=torch.float32, device=cif.device)
        ids = torch.empty((B, self.max_annotations), dtype=torch.int64, device=cif.device)
        counts = torch.empty((B,), dtype=torch.int32, device=cif.device)
        n_initial = 0
        init_t = ids_t = None
        if initial_annotations is not None and initial_annotations.shape[-3] > 0:
            init_t, _ = _prep(initial_annotations)
            init_t = init_t.reshape(B, -1, K, 4)
            n_initial = init_t.shape[1]
            if initial_ids is None:
                raise RuntimeError('require initial_ids when initial_annotations are given')   # cifcaf.cpp:178
            ids_t, _ = _prep(initial_ids, torch.int64)
            ids_t = ids_t.reshape(B, n_initial)
        _lib.check(_lib.lib().opa_cifcaf_decode(
            self._handle, ctypes.byref(shape), ctypes.byref(params) if params is not None else None,
            _ptr(cif), _ptr(caf), _ptr(init_t), _ptr(ids_t), n_initial,
            _ptr(ws), ws.numel(), _ptr(out), _ptr(ids), _ptr(counts), _stream()), 'opa_cifcaf_decode')
        self._last = (shape, ws)
        if orig.type != 'cuda':
            out, ids, counts = out.to(orig), ids.to(orig), counts.to(orig)
        return out, ids, counts

    def capture(self, cif, cif_stride, caf, caf_stride, *, params=None, stream=None):
        """Capture one ``call_batch`` on the given (static) field tensors as a HIP graph: the whole decode
        -- ten kernels, two memsets, no allocation inside the C call -- then costs one ``graph.replay()``
        instead of a dozen launches from Python.  Returns ``(graph, (annotations, ids, counts))``; refill
        ``cif`` / ``caf`` in place and replay.  Independent decoders on separate streams run concurrently
        (a batch-32 association occupies 32 of the 256 CUs)."""
        stream = stream or torch.cuda.Stream(device=cif.device)
        with torch.cuda.stream(stream):
            self.call_batch(cif, cif_stride, caf, caf_stride, params=params)     # workspace allocated, header valid
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = self.call_batch(cif, cif_stride, caf, caf_stride, params=params)
        # the graph replays into this workspace: keep it alive with the graph and refuse to swap it
        graph._opa_keepalive = (self, self._last[1], cif, caf)
        self._pinned = True
        return graph, out

    def call_with_initial_annotations(self, cif_field, cif_stride, caf_field, caf_stride,
                                      initial_annotations=None, initial_ids=None):
        """Single image, like module.cpp:36: ``-> (Tensor[n,K,4] (v,x,y,s), Tensor[n] int64)``."""
        ia = initial_annotations.unsqueeze(0) if initial_annotations is not None else None
        ii = initial_ids.unsqueeze(0) if initial_ids is not None else None
        out, ids, counts = self.call_batch(cif_field.unsqueeze(0), cif_stride, caf_field.unsqueeze(0),
                                           caf_stride, ia, ii)
        n = int(counts[0])
        if n & COUNT_FAILED and self.pool_overflowed():          # the map did not fit the automatic pool: once more, full pool
            self.use_full_pool()
            out, ids, counts = self.call_batch(cif_field.unsqueeze(0), cif_stride, caf_field.unsqueeze(0),
                                               caf_stride, ia, ii)
            n = int(counts[0])
        check_counts(n)
        if n & COUNT_OVERFLOW:
            raise _lib.NativeError('annotation capacity overflow: %d poses dropped; construct CifCaf with a '
                                   'larger max_annotations' % int(self.workspace_view('status', torch.int32)[0]))
        n &= COUNT_ROWS_MASK
        return out[0, :n].clone(), ids[0, :n].clone()

    def call(self, cif_field, cif_stride, caf_field, caf_stride):
        """module.cpp:35."""
        return self.call_with_initial_annotations(cif_field, cif_stride, caf_field, caf_stride)

    def workspace_view(self, what, dtype=torch.uint8):
        """Flat tensor view of an intermediate buffer of the last ``call_batch`` (debug/tests)."""
        shape, ws = self._last
        off, size = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(_lib.lib().opa_cifcaf_workspace_view(ctypes.byref(shape), what.encode(), ctypes.byref(off),
                                                        ctypes.byref(size)), 'opa_cifcaf_workspace_view')
        return ws[off.value:off.value + size.value].view(dtype)

    def assoc_stats(self):
        """Statistics of the last ``call_batch``'s association kernel: int32 ``[B,24]`` (see
        ``opa_cifcaf_workspace_view`` in the header): growths started / accepted / cancelled / dropped,
        mispredictions, ticks."""
        shape, _ = self._last
        return self.workspace_view('assoc_stats', torch.int32)[:shape.batch * 24].view(shape.batch, 24)   # (regions are padded to 256 B)

    def get_cifhr(self, image=0):
        """module.cpp:37-39 -> (Tensor [F,Hhr,Whr], revision).  The decode keeps the map as a pool of tiles; this gathers the
        dense array the reference returns (0.0 = untouched, else 1 + value) for one image of the last ``call_batch``."""
        if self._last is None:
            return torch.zeros((1, 1, 1)), 0.0
        shape, ws = self._last
        off, rows, cols, pitch = ctypes.c_size_t(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        rev = ctypes.c_double()
        _lib.check(_lib.lib().opa_cifcaf_cifhr_view(ctypes.byref(shape), ctypes.byref(off), ctypes.byref(rows),
                                                    ctypes.byref(cols), ctypes.byref(pitch), ctypes.byref(rev)))
        dense = torch.empty((shape.n_cif, rows.value, cols.value), dtype=torch.float32, device=ws.device)
        _lib.check(_lib.lib().opa_cifcaf_get_cifhr(ctypes.byref(shape), _ptr(ws), int(image), _ptr(dense), _stream()),
                   'opa_cifcaf_get_cifhr')
        return dense, rev.value


_hw_queues_checked = False


def ensure_hw_queues(lanes):
    """Decode lanes (one HIP stream each) only run side by side while every stream has a hardware queue of its own; the HIP
    runtime maps all streams of a process onto FOUR by default, and four lanes then share queues with the network's stream
    (round 4: 54 k images/s with four lanes against 66 k with two; with eight queues 86 k).  The runtime reads
    ``GPU_MAX_HW_QUEUES`` when it initialises, i.e. at the first GPU call of the process.  Called where lanes are set up
    (:class:`DecodeLanes`, ``Predictor``): sets the variable to 8 when that is still possible and nothing was chosen -- an
    explicit setting wins, importing the package changes nothing -- and says so, once, when it is too late."""
    global _hw_queues_checked
    if lanes <= 2 or _hw_queues_checked or os.environ.get('GPU_MAX_HW_QUEUES'):
        return
    _hw_queues_checked = True
    if torch.cuda.is_initialized():
        import warnings
        warnings.warn('openpifpaf_amd: %d decode lanes on the HIP runtime\'s default of 4 hardware queues -- they will share '
                      'queues; export GPU_MAX_HW_QUEUES=8 before the first GPU call of the process' % lanes, RuntimeWarning)
    else:
        os.environ['GPU_MAX_HW_QUEUES'] = '8'


class DecodeLanes:
    """Several batched decodes in flight at once.  One decode is six kernels in a row whose longest, the
    association, keeps one workgroup per image busy (32 of 256 compute units for a batch of 32) -- the stages of the
    NEXT batch fit beside it.  Each lane is a :class:`CifCaf` of its own (workspace) on a stream of its own;
    ``submit`` round-robins over them and returns a ticket instead of making the caller's stream wait::

        lanes = DecodeLanes(17, skeleton, lanes=2)
        tickets = [lanes.submit(cif_i, 8, caf_i, 8) for ...]      # field tensors stay valid until the ticket is done
        out, ids, counts = tickets[0].result()                     # current stream waits for that decode only
    """

    class Ticket:
        def __init__(self, tensors, event):
            self.tensors, self.event = tensors, event

        def result(self):
            """The decode's ``(annotations, ids, counts)``, safe to use on the current stream."""
            consumer = torch.cuda.current_stream()
            consumer.wait_event(self.event)
            for t in self.tensors:                       # allocated on the lane's stream: the allocator must not hand
                if t.is_cuda:                            # them to the lane's next submit while the consumer still reads
                    t.record_stream(consumer)
            return self.tensors

        def synchronize(self):
            self.event.synchronize()
            return self.tensors

    def __init__(self, n_keypoints, skeleton, *, lanes=2, max_annotations=DEFAULT_MAX_ANNOTATIONS, cifhr_pool_tiles=0):
        ensure_hw_queues(lanes)
        self.decoders = [CifCaf(n_keypoints, skeleton, max_annotations=max_annotations, cifhr_pool_tiles=cifhr_pool_tiles)
                         for _ in range(max(1, lanes))]
        if len(self.decoders) > 1:                       # the tie pass inside the association kernel: +11 % with twelve lanes (round 4)
            for d in self.decoders:
                d.set_tie_placement(True)
        self.streams = [torch.cuda.Stream(priority=-1) for _ in self.decoders]
        self._next = 0

    def set_debug(self, **switches):
        """``CifCaf.set_debug`` on every lane's decoder."""
        for d in self.decoders:
            d.set_debug(**switches)

    def submit(self, cif, cif_stride, caf, caf_stride, *, params=None):
        lane = self._next
        self._next = (self._next + 1) % len(self.decoders)
        stream = self.streams[lane]
        ready = torch.cuda.Event()
        ready.record()                                   # the fields are complete on the caller's stream
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            tensors = self.decoders[lane].call_batch(cif, cif_stride, caf, caf_stride, params=params)
            for t in (cif, caf):
                if t.is_cuda:
                    t.record_stream(stream)              # the allocator must not hand the fields out before the lane is done
            done = torch.cuda.Event()
            done.record(stream)
        return DecodeLanes.Ticket(tensors, done)


def grow_connection_blend(caf, x, y, s, filter_sigmas=1.0, only_max=False):
    """``torch.ops.openpifpaf_decoder.grow_connection_blend`` (module.cpp:55) -> [x, y, s, v]."""
    caf, _ = _prep(caf)
    caf = caf.reshape(-1, 7)
    out = (ctypes.c_double * 4)()
    _lib.check(_lib.lib().opa_grow_connection_blend(_ptr(caf) if caf.numel() else None, caf.shape[0],
                                                    float(x), float(y), float(s), float(filter_sigmas),
                                                    int(bool(only_max)), out, _stream()),
               'opa_grow_connection_blend')
    return list(out)


class CifHr:
    """``torch.classes.openpifpaf_decoder_utils.CifHr`` (module.cpp:75-84), batched."""
    get_neighbors, set_neighbors = _static_getset('cifhr_neighbors', int)
    get_threshold, set_threshold = _static_getset('cif_threshold')
    get_ablation_skip, set_ablation_skip = _static_getset('ablation_cifhr_skip', bool)

    def __init__(self):
        self.accumulated = None     # [B,F,rows,pitch]
        self.cols = 0
        self.revision = 0.0

    def accumulate(self, cif_field, stride, min_scale=0.0, factor=1.0, *, params=None):
        """reset() + accumulate() of a fresh instance.  ``cif_field``: [F,5,H,W] or [B,F,5,H,W]."""
        cif, _ = _prep(cif_field)
        if cif.dim() == 4:
            cif = cif.unsqueeze(0)
        B, F, _, H, W = cif.shape
        L = _lib.lib()
        pitch = L.opa_cifhr_pitch(W, int(stride))
        rows, self.cols = (H - 1) * stride + 1, (W - 1) * stride + 1
        self.accumulated = torch.empty((B, F, rows, pitch), dtype=torch.float32, device=cif.device)
        nbytes = L.opa_cifhr_scratch_bytes(B, F, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=cif.device)
        _lib.check(L.opa_cifhr_accumulate(_ptr(cif), B, F, H, W, int(stride), float(min_scale), float(factor),
                                          ctypes.byref(params) if params is not None else None,
                                          _ptr(self.accumulated), _ptr(scratch), nbytes, _stream()),
                   'opa_cifhr_accumulate')
        self.revision = 1.0
        self._scratch = scratch     # keep alive until the stream has run

    def get_accumulated(self, image=0):
        return self.accumulated[image, :, :, :self.cols], self.revision


class CifSeeds:
    """``torch.classes.openpifpaf_decoder_utils.CifSeeds`` (module.cpp:86-94), batched."""
    get_threshold, set_threshold = _static_getset('seed_threshold')
    get_ablation_nms, set_ablation_nms = _static_getset('ablation_cifseeds_nms', bool)
    get_ablation_no_rescore, set_ablation_no_rescore = _static_getset('ablation_cifseeds_no_rescore', bool)

    def __init__(self, cifhr: CifHr):
        self.cifhr = cifhr
        self._out = None

    def fill(self, cif_field, stride, *, params=None):
        cif, _ = _prep(cif_field)
        if cif.dim() == 4:
            cif = cif.unsqueeze(0)
        B, F, _, H, W = cif.shape
        cap = F * H * W
        L = _lib.lib()
        f = torch.empty((B, cap), dtype=torch.int32, device=cif.device)
        vxys = torch.empty((B, cap, 4), dtype=torch.float32, device=cif.device)
        count = torch.empty((B,), dtype=torch.int32, device=cif.device)
        nbytes = L.opa_cifseeds_scratch_bytes(B, F, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=cif.device)
        _lib.check(L.opa_cifseeds_fill(_ptr(cif), B, F, H, W, int(stride), _ptr(self.cifhr.accumulated),
                                       ctypes.byref(params) if params is not None else None,
                                       _ptr(f), _ptr(vxys), _ptr(count), _ptr(scratch), nbytes, _stream()),
                   'opa_cifseeds_fill')
        self._out = (f, vxys, count, scratch)

    def get(self, image=0):
        """-> (fields int64 [n], vxys float32 [n,4]) sorted by v descending (cif_seeds.cpp:93-114)."""
        f, vxys, count, _ = self._out
        n = int(count[image])
        return f[image, :n].to(torch.int64), vxys[image, :n]


class CifDetSeeds:
    """``torch.classes.openpifpaf_decoder_utils.CifDetSeeds`` (module.cpp:96-102).  ``cifhr``: the detection
    map ``[F, rows, cols]`` at revision 1.0 (any device; it is re-pitched on the GPU)."""
    get_threshold, set_threshold = _static_getset('seed_threshold')

    def __init__(self, cifhr, revision=1.0):
        if revision != 1.0:
            raise RuntimeError('the HIP path stores the map at revision 1.0 (a fresh reference instance)')
        hr, _ = _prep(cifhr)
        F, rows, cols = hr.shape
        pitch = _lib.lib().opa_cifhr_pitch(cols, 1)
        self.accumulated = torch.zeros((F, rows, pitch), dtype=torch.float32, device=hr.device)
        self.accumulated[:, :, :cols] = hr
        self.cols = cols
        self._out = None

    def fill(self, field, stride, *, params=None):
        field, _ = _prep(field)
        F, C, H, W = field.shape
        if C != 6 or tuple(self.accumulated.shape[:2]) != (F, (H - 1) * stride + 1) or self.cols != (W - 1) * stride + 1:
            raise RuntimeError('expected a CifDet field [F,6,H,W] matching the map')
        cap = F * H * W
        L = _lib.lib()
        f = torch.empty((cap,), dtype=torch.int32, device=field.device)
        vxywh = torch.empty((cap, 5), dtype=torch.float32, device=field.device)
        count = torch.empty((1,), dtype=torch.int32, device=field.device)
        nbytes = L.opa_cifseeds_scratch_bytes(1, F, H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=field.device)
        _lib.check(L.opa_cifdetseeds_fill(_ptr(field), 1, F, H, W, int(stride), _ptr(self.accumulated),
                                          ctypes.byref(params) if params is not None else None,
                                          _ptr(f), _ptr(vxywh), _ptr(count), _ptr(scratch), nbytes, _stream()),
                   'opa_cifdetseeds_fill')
        self._out = (f, vxywh, count, scratch)

    def get(self):
        """-> (fields int64 [n], vxywh float32 [n,5]) sorted by v descending (cif_seeds.cpp:117-139)."""
        f, vxywh, count, _ = self._out
        n = int(count[0])
        return f[:n].to(torch.int64), vxywh[:n]


class CafScored:
    """``torch.classes.openpifpaf_decoder_utils.CafScored`` (module.cpp:104-111), batched."""
    get_default_score_th, set_default_score_th = _static_getset('caf_threshold')
    get_ablation_no_rescore, set_ablation_no_rescore = _static_getset('ablation_caf_no_rescore', bool)

    def __init__(self, cifhr: CifHr, cif_shape, cif_stride, score_th=-1.0, cif_floor=0.1):
        self.cifhr = cifhr
        self.cif_shape = tuple(cif_shape)[-4:]     # (F,5,H,W)
        self.cif_stride = int(cif_stride)
        self.score_th = float(score_th)
        self.cif_floor = float(cif_floor)
        self._out = None

    def fill(self, caf_field, stride, skeleton, *, params=None):
        caf, _ = _prep(caf_field)
        if caf.dim() == 4:
            caf = caf.unsqueeze(0)
        B, A, _, H, W = caf.shape
        F, _, cH, cW = self.cif_shape
        skel, _ = _prep(torch.as_tensor(skeleton), torch.int64)
        lists = torch.empty((B, A, 2, 7, H * W), dtype=torch.float32, device=caf.device)
        counts = torch.empty((B, A, 2), dtype=torch.int32, device=caf.device)
        _lib.check(_lib.lib().opa_cafscored_fill(
            _ptr(caf), B, A, H, W, int(stride), _ptr(self.cifhr.accumulated), F, cH, cW, self.cif_stride,
            _ptr(skel), self.score_th, self.cif_floor, ctypes.byref(params) if params is not None else None,
            _ptr(lists), _ptr(counts), _stream()), 'opa_cafscored_fill')
        self._out = (lists, counts, skel)

    def get(self, image=0):
        """-> (forward, backward): lists of [n,7] tensors (caf_scored.cpp:86-104)."""
        lists, counts, _ = self._out
        counts = counts[image].cpu()
        fwd = [lists[image, a, 0, :, :int(counts[a, 0])].t().contiguous() for a in range(lists.shape[1])]
        bwd = [lists[image, a, 1, :, :int(counts[a, 1])].t().contiguous() for a in range(lists.shape[1])]
        return fwd, bwd


class NMSKeypoints:
    """Static tunables of ``openpifpaf_decoder_utils.NMSKeypoints`` (module.cpp:113-117)."""
    get_instance_threshold, set_instance_threshold = _static_getset('nms_instance_threshold')
    get_keypoint_threshold, set_keypoint_threshold = _static_getset('nms_keypoint_threshold')
    get_suppression, set_suppression = _static_getset('nms_suppression')


class CifDet:
    """Drop-in for ``torch.classes.openpifpaf_decoder.CifDet`` (module.cpp:57-62), batched."""
    _max_detections_before_nms = 120                   # cifdet.cpp:16

    @classmethod
    def get_max_detections_before_nms(cls):
        return cls._max_detections_before_nms

    @classmethod
    def set_max_detections_before_nms(cls, v):
        cls._max_detections_before_nms = int(v)

    def __init__(self):
        _device()
        self._workspaces = {}

    def call_batch(self, cifdet_field, cifdet_stride, *, params=None):
        """``cifdet_field`` [B,F,6,H,W] -> device tensors (categories [B,max] int64, scores [B,max],
        boxes [B,max,4] (x0,y0,x1,y1), counts [B] int32)."""
        field, orig = _prep(cifdet_field)
        B, F, C, H, W = field.shape
        if C != 6:
            raise ValueError('expected a CifDet field [B,F,6,H,W], got %s' % (tuple(field.shape),))
        shape = _lib.DetShape(B, F, H, W, int(cifdet_stride), self._max_detections_before_nms)
        key = (B, F, H, W, int(cifdet_stride), shape.max_detections, field.device.index)
        ws = self._workspaces.get(key)
        if ws is None:
            nbytes = _lib.lib().opa_cifdet_workspace_bytes(ctypes.byref(shape))
            if nbytes == 0:
                raise _lib.NativeError(_lib.lib().opa_last_error().decode())
            self._workspaces.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=field.device)
            self._workspaces[key] = ws
        M = shape.max_detections
        cat = torch.empty((B, M), dtype=torch.int64, device=field.device)
        sc = torch.empty((B, M), dtype=torch.float32, device=field.device)
        bx = torch.empty((B, M, 4), dtype=torch.float32, device=field.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=field.device)
        _lib.check(_lib.lib().opa_cifdet_decode(
            ctypes.byref(shape), ctypes.byref(params) if params is not None else None, _ptr(field),
            _ptr(ws), ws.numel(), _ptr(cat), _ptr(sc), _ptr(bx), _ptr(cnt), _stream()), 'opa_cifdet_decode')
        if orig.type != 'cuda':
            cat, sc, bx, cnt = cat.to(orig), sc.to(orig), bx.to(orig), cnt.to(orig)
        return cat, sc, bx, cnt

    def call(self, cifdet_field, cifdet_stride):
        """Single image, like module.cpp:61: -> (categories [n], scores [n], boxes [n,4])."""
        cat, sc, bx, cnt = self.call_batch(cifdet_field.unsqueeze(0), cifdet_stride)
        n = int(cnt[0])
        return cat[0, :n].clone(), sc[0, :n].clone(), bx[0, :n].clone()
