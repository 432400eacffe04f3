"""Decoder plugin layer: the reference's ``openpifpaf.decoder`` interface for the
CifCaf path, backed by the HIP library.

Mirrors (same names, argument meaning and error behaviour):

* ``Decoder`` base -- reference ``decoder/decoder.py:21-145`` (``cli``, ``configure``,
  ``factory``, ``__call__``, ``fields_batch``, ``batch``, ``last_nn_time`` /
  ``last_decoder_time``, pickling without the worker pool).
* ``CifCaf`` -- reference ``decoder/cifcaf.py:81-277``.  ``__call__(fields)`` takes the
  per-image list of head tensors; ``batch(model, image_batch)`` is overridden to keep
  the head outputs ON THE DEVICE and decode the whole batch with one native call
  (the reference copies every head to the host, ``decoder.py:96-100``, and decodes
  image by image, optionally in a fork pool, ``decoder.py:130-131``).
* ``DECODERS`` / ``factory`` / ``cli`` / ``configure`` / ``Multi`` -- reference
  ``decoder/factory.py:17-172`` and ``decoder/multi.py:11-35``.
* ``register()`` (package level) adds ``CifCaf`` to a host ``openpifpaf.DECODERS`` so
  that the reference's ``--decoder=cifcaf:0`` selects it (plugin discovery,
  reference ``plugin.py:17-40``: this package's name starts with ``openpifpaf_``).
"""
import argparse
import logging
import time
from collections import defaultdict
from typing import List, Optional

import numpy as np
import torch

from . import headmeta, native
from .annotation import Annotation, AnnotationDet

LOG = logging.getLogger(__name__)


class Decoder:
    """Generate predictions from image or field inputs (reference ``decoder.py:21``)."""
    default_worker_pool = None
    torch_decoder = True

    def __init__(self):
        self.priority = 0.0
        self.worker_pool = None      # the device decodes whole batches; no fork pool needed
        self.last_decoder_time = 0.0
        self.last_nn_time = 0.0

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        """Command line interface (CLI) to extend argument parser."""

    @classmethod
    def configure(cls, args: argparse.Namespace):
        """Take the parsed argument parser output and configure class variables."""

    @classmethod
    def factory(cls, head_metas) -> List['Decoder']:
        raise NotImplementedError

    def __call__(self, fields, *, initial_annotations=None) -> List[Annotation]:
        raise NotImplementedError

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ('worker_pool',)}

    @classmethod
    def fields_batch(cls, model, image_batch, *, device=None, to_cpu=False):
        """From image batch to per-image field lists.  Unlike the reference
        (``decoder.py:76-112``) the heads stay on the device unless ``to_cpu``."""
        with torch.no_grad():
            if device is not None:
                image_batch = image_batch.to(device, non_blocking=True)
            heads = model(image_batch)
            if to_cpu:
                heads = [h.cpu() for h in heads]
        return [[h[b] for h in heads] for b in range(len(image_batch))]

    def batch(self, model, image_batch, *, device=None, gt_anns_batch=None):
        """From image batch straight to annotations batch (reference ``decoder.py:114-137``)."""
        start_nn = time.perf_counter()
        fields_batch = self.fields_batch(model, image_batch, device=device)
        self.last_nn_time = time.perf_counter() - start_nn
        start_decoder = time.perf_counter()
        result = [self(fields) for fields in fields_batch]
        self.last_decoder_time = time.perf_counter() - start_decoder
        return result


class CifCaf(Decoder):
    """Generate CifCaf poses from fields on the MI355X."""
    connection_method = 'blend'
    nms_before_force_complete = False
    reverse_match = True
    max_annotations = native.DEFAULT_MAX_ANNOTATIONS
    #: the reference's class carries its occupancy visualizer here (``decoder/cifcaf.py:87``) and its TrackingPose looks at it
    #: after the soft NMS (``tracking_pose.py:160``); the occupancy of this decode is a bitmap on the device: none
    occupancy_visualizer = None
    #: per-image capacity of the high-resolution map's tile pool (``opa_shape::cifhr_pool_tiles``): 0 = automatic,
    #: ``'full'`` / -1 = every tile (can never run out), n = n tiles; ``--cifcaf-cifhr-pool-tiles``
    cifhr_pool_tiles = 0

    def __init__(self, cif_metas: List[headmeta.Cif], caf_metas: List[headmeta.Caf]):
        super().__init__()
        self.cif_metas = cif_metas
        self.caf_metas = caf_metas
        self.score_weights = cif_metas[0].score_weights
        self.confidence_scales = caf_metas[0].decoder_confidence_scales
        self.cpp_decoder = native.CifCaf(
            len(cif_metas[0].keypoints),
            torch.LongTensor(caf_metas[0].skeleton) - 1,          # reference cifcaf.py:119-122
            max_annotations=self.max_annotations,
            cifhr_pool_tiles=self.cifhr_pool_tiles,
        )
        # prefer decoders with more keypoints and associations (reference cifcaf.py:123-125)
        self.priority += sum(m.n_fields for m in cif_metas) / 1000.0
        self.priority += sum(m.n_fields for m in caf_metas) / 1000.0

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        """Reference ``cifcaf.py:127-172``."""
        C = native.CifCaf
        group = parser.add_argument_group('CifCaf decoder')
        group.add_argument('--force-complete-pose', default=False, action='store_true')
        group.add_argument('--force-complete-caf-th', type=float, default=C.get_force_complete_caf_th(),
                           help='CAF threshold for force complete. Set to -1 to deactivate.')
        group.add_argument('--nms-before-force-complete', default=False, action='store_true',
                           help='run an additional NMS before completing poses')
        group.add_argument('--keypoint-threshold', type=float, default=C.get_keypoint_threshold(),
                           help='filter keypoints by score')
        group.add_argument('--keypoint-threshold-rel', type=float, default=C.get_keypoint_threshold_rel(),
                           help='filter keypoint connections by relative score')
        group.add_argument('--greedy', default=False, action='store_true', help='greedy decoding')
        group.add_argument('--connection-method', default=cls.connection_method, choices=('max', 'blend'),
                           help='connection method to use, max is faster')
        group.add_argument('--cifcaf-block-joints', default=False, action='store_true', help='block joints')
        group.add_argument('--no-reverse-match', default=True, dest='reverse_match', action='store_false')
        group.add_argument('--ablation-cifseeds-nms', default=False, action='store_true')
        group.add_argument('--ablation-cifseeds-no-rescore', default=False, action='store_true')
        group.add_argument('--ablation-caf-no-rescore', default=False, action='store_true')
        group.add_argument('--ablation-independent-kp', default=False, action='store_true')
        group.add_argument('--cifcaf-max-annotations', default=cls.max_annotations, type=int,
                           help='per-image capacity of the device-side annotation buffer')
        group.add_argument('--cifcaf-cifhr-pool-tiles', default=cls.cifhr_pool_tiles,
                           type=lambda v: -1 if v == 'full' else int(v),
                           help='32x64 tiles per image of the high-resolution map\'s pool: 0 = automatic, full = every tile')

    @classmethod
    def configure(cls, args: argparse.Namespace):
        """Reference ``cifcaf.py:174-211``: CLI -> the native library's global tunables."""
        C = native.CifCaf
        keypoint_threshold_nms = args.keypoint_threshold
        if args.force_complete_pose:
            if not args.ablation_independent_kp:
                args.keypoint_threshold = 0.0
            args.keypoint_threshold_rel = 0.0
            keypoint_threshold_nms = 0.0
        if args.seed_threshold < args.keypoint_threshold:
            LOG.warning('consistency: decreasing keypoint threshold to seed threshold of %f',
                        args.seed_threshold)
            args.keypoint_threshold = args.seed_threshold

        cls.nms_before_force_complete = args.nms_before_force_complete
        native.NMSKeypoints.set_keypoint_threshold(keypoint_threshold_nms)
        C.set_force_complete(args.force_complete_pose)
        C.set_force_complete_caf_th(args.force_complete_caf_th)
        C.set_keypoint_threshold(args.keypoint_threshold)
        C.set_keypoint_threshold_rel(args.keypoint_threshold_rel)
        C.set_greedy(args.greedy)
        C.set_block_joints(args.cifcaf_block_joints)
        cls.connection_method = args.connection_method
        cls.reverse_match = args.reverse_match
        C.set_reverse_match(args.reverse_match)
        native.CifSeeds.set_ablation_nms(args.ablation_cifseeds_nms)
        native.CifSeeds.set_ablation_no_rescore(args.ablation_cifseeds_no_rescore)
        native.CafScored.set_ablation_no_rescore(args.ablation_caf_no_rescore)
        if args.ablation_cifseeds_no_rescore and args.ablation_caf_no_rescore:
            native.CifHr.set_ablation_skip(True)
        cls.max_annotations = getattr(args, 'cifcaf_max_annotations', cls.max_annotations)
        cls.cifhr_pool_tiles = getattr(args, 'cifcaf_cifhr_pool_tiles', cls.cifhr_pool_tiles)

    @classmethod
    def factory(cls, head_metas):
        """Reference ``cifcaf.py:213-222``."""
        return [
            cls([meta], [meta_next])
            for meta, meta_next in zip(head_metas[:-1], head_metas[1:])
            if isinstance(meta, headmeta.Cif) and isinstance(meta_next, headmeta.Caf)
        ]

    # ---- tensors <-> Annotation objects -------------------------------------------
    def _annotations_py(self, ann_data: np.ndarray, ann_ids: np.ndarray) -> List[Annotation]:
        """Reference ``cifcaf.py:262-272``: native (v,x,y,s) rows -> Annotation (x,y,v) + joint_scales."""
        out = []
        for data, ann_id in zip(ann_data, ann_ids):
            ann = Annotation(self.cif_metas[0].keypoints, self.caf_metas[0].skeleton,
                             score_weights=self.score_weights)
            ann.data[:, :2] = data[:, 1:3]
            ann.data[:, 2] = data[:, 0]
            ann.joint_scales[:] = data[:, 3]
            if ann_id != -1:
                ann.id_ = int(ann_id)
            out.append(ann)
        return out

    @staticmethod
    def _initial_tensors(initial_annotations, n_fields):
        """Reference ``cifcaf.py:225-239``."""
        if not initial_annotations:
            return None, None
        t = torch.empty((len(initial_annotations), n_fields, 4))
        ids = torch.empty((len(initial_annotations),), dtype=torch.int64)
        for i, ann_py in enumerate(initial_annotations):
            for f in range(len(ann_py.data)):
                t[i, f, 0] = float(ann_py.data[f, 2])
                t[i, f, 1] = float(ann_py.data[f, 0])
                t[i, f, 2] = float(ann_py.data[f, 1])
                t[i, f, 3] = float(ann_py.joint_scales[f])
            ids[i] = getattr(ann_py, 'id_', -1)
        return t, ids

    def __call__(self, fields, initial_annotations=None):
        """Single image: ``fields[meta.head_index]`` are ``[F,C,H,W]`` tensors (device or CPU)."""
        init_t, ids_t = self._initial_tensors(initial_annotations, self.cif_metas[0].n_fields)
        start = time.perf_counter()
        annotations, annotation_ids = self.cpp_decoder.call_with_initial_annotations(
            fields[self.cif_metas[0].head_index], self.cif_metas[0].stride,
            fields[self.caf_metas[0].head_index], self.caf_metas[0].stride,
            init_t, ids_t)
        annotations = annotations.cpu().numpy()
        annotation_ids = annotation_ids.cpu().numpy()
        LOG.debug('native annotations = %d (%.1fms)', len(annotations), (time.perf_counter() - start) * 1000.0)
        return self._annotations_py(annotations, annotation_ids)

    def decode_heads(self, heads):
        """Batched device decode of head outputs ``heads[head_index] = [B,F,C,H,W]``.
        Returns device tensors ``(annotations [B,max,K,4], ids [B,max], counts [B])``; asynchronous."""
        return self.cpp_decoder.call_batch(
            heads[self.cif_metas[0].head_index], self.cif_metas[0].stride,
            heads[self.caf_metas[0].head_index], self.caf_metas[0].stride)

    supports_device_inverse = True       # batch(..., meta_batch=...) undoes pad / rescale / flip on the device

    # ---- several batches in flight (the reference's --decoder-workers, decoder/decoder.py:33-47,130-131) ----------
    #: decode lanes: decoders / workspaces / HIP streams taking batches in turn.  The reference overlaps the decode of a
    #: batch with the next batch's network through a fork pool of CPU workers; here a lane's decode (whose longest
    #: kernel keeps one workgroup per image busy) runs on its own stream beside the next batch's network, and the
    #: annotations come back through pinned memory without stalling either stream.
    decoder_workers = 2

    def set_debug(self, **switches):
        """A/B and test switches (``opa_debug``: exact kernel variants, the watchdog, measurements; none changes a result) of
        this decoder's native handles -- the synchronous one and every decode lane's.  ``set_debug()`` restores the defaults."""
        self._debug = dict(switches)
        self.cpp_decoder.set_debug(**switches)
        if getattr(self, '_lanes', None) is not None:
            self._lanes.set_debug(**switches)

    def _decode_lanes(self):
        lanes = getattr(self, '_lanes', None)
        n = max(1, int(self.decoder_workers or 1))
        key = (n, int(self.max_annotations), self.cifhr_pool_tiles)
        if lanes is None or getattr(self, '_lanes_key', None) != key:
            # (the lanes, their workspaces and the pinned host buffers are sized by these: rebuilt when one of them changes;
            # batches still in flight on the old lanes are collected first)
            for pending in list(getattr(self, '_lane_pending', {}).values()):
                pending.result()
            lanes = native.DecodeLanes(len(self.cif_metas[0].keypoints), torch.LongTensor(self.caf_metas[0].skeleton) - 1,
                                       lanes=n, max_annotations=self.max_annotations, cifhr_pool_tiles=self.cifhr_pool_tiles)
            if getattr(self, '_debug', None):
                lanes.set_debug(**self._debug)
            self._lanes, self._lane_host, self._lane_pending, self._lanes_key = lanes, {}, {}, key
        return lanes

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ('worker_pool', '_lanes', '_lane_host', '_lane_pending', '_lanes_key')}

    class Pending:
        """A batch in flight: ``result()`` waits for ITS decode only (an event on its lane) and builds the annotations."""

        def __init__(self, owner, lane, event, host, n_images, t_submit, fields=None):
            self.owner, self.lane, self.event, self.host, self.n_images, self.t_submit = owner, lane, event, host, n_images, t_submit
            self.fields = fields                 # (cif, caf, meta_batch): kept for the one case a decode is repeated (below)
            self._result = None
            self._error = None

        def done(self):
            return self._result is not None or self.event.query()

        def result(self):
            if self._error is not None:
                raise self._error                # (the decode failed: every call says so, the lane itself is free again)
            if self._result is None:
                try:
                    self.event.synchronize()
                    out, ids, counts = (t.numpy() for t in self.host)
                    if (counts[:self.n_images] & native.COUNT_FAILED).any() and self.fields is not None and \
                            self.owner._lanes.decoders[self.lane].pool_overflowed():
                        # an image's CIF map did not fit the tile pool and the batch's spill region (several structureless,
                        # all-active fields in one batch): from now on every lane uses a pool that holds the whole map, and
                        # this batch is decoded again, synchronously
                        out, ids, counts = self.owner._decode_again_with_full_pool(*self.fields)
                    self._result = self.owner._annotations_from_host(out, ids, counts[:self.n_images])
                    self.owner.last_decoder_time = time.perf_counter() - self.t_submit
                except Exception as e:           # noqa: BLE001 -- remembered for later calls, raised below
                    self._error = e
                    raise
                finally:
                    # whatever happened, the ticket is spent: a failed batch must not be collected again before every
                    # later submit to this lane (and fail every later, unrelated batch)
                    self.fields = None
                    if self.owner._lane_pending.get(self.lane) is self:
                        del self.owner._lane_pending[self.lane]
            return self._result

    def _decode_again_with_full_pool(self, cif, caf, meta_batch):
        for dec in [self.cpp_decoder] + (list(self._lanes.decoders) if getattr(self, '_lanes', None) is not None else []):
            if dec.cifhr_pool_tiles != -1:
                dec.use_full_pool()
        LOG.warning('a CIF map reached more tiles than the automatic pool holds: decoding again with a full pool '
                    '(--cifcaf-cifhr-pool-tiles full, or decoder.CifCaf.cifhr_pool_tiles = -1, avoids the second decode)')
        out, ids, counts = self.cpp_decoder.call_batch(cif, self.cif_metas[0].stride, caf, self.caf_metas[0].stride)
        if meta_batch is not None:
            from .annotation import inverse_transform_batch
            out = inverse_transform_batch(out, meta_batch)
        return out.cpu().numpy(), ids.cpu().numpy(), counts.cpu().numpy()

    def _annotations_from_host(self, out, ids, counts):
        native.check_counts(counts)                 # a watchdog failure raises instead of decoding to "no poses"
        result = []
        for b in range(len(counts)):
            n = int(counts[b]) & native.COUNT_ROWS_MASK           # valid rows
            if int(counts[b]) & native.COUNT_OVERFLOW:
                LOG.warning('image %d: annotations dropped for lack of capacity (raise --cifcaf-max-annotations)', b)
            result.append(self._annotations_py(out[b, :n], ids[b, :n]))
        return result

    def batch_async(self, model, image_batch, *, device=None, meta_batch=None):
        """Like :meth:`batch`, but returns at once with a :class:`Pending`: the network is queued on the current
        stream, the decode (+ the inverse transform with ``meta_batch``) on the next lane's stream behind it, the
        annotations travel to pinned host memory on that stream.  Submit batch *i+1* before asking for batch *i*'s
        ``result()`` and the two overlap; up to ``decoder_workers`` batches may be in flight (submitting to a lane
        whose previous batch was not collected yet collects it first: host buffers are per lane)."""
        t0 = time.perf_counter()
        lanes = self._decode_lanes()
        with torch.no_grad():
            if device is not None:
                image_batch = image_batch.to(device, non_blocking=True)
            heads = model(image_batch)
        cif, caf = heads[self.cif_metas[0].head_index], heads[self.caf_metas[0].head_index]
        lane = lanes._next
        stale = self._lane_pending.get(lane)
        if stale is not None:
            stale.result()
        ticket = lanes.submit(cif, self.cif_metas[0].stride, caf, self.caf_metas[0].stride)
        stream = lanes.streams[lane]
        B = cif.shape[0]
        key = (lane, B)
        host = self._lane_host.get(key)
        if host is None:
            K = len(self.cif_metas[0].keypoints)
            host = (torch.empty((B, self.max_annotations, K, 4), dtype=torch.float32).pin_memory(),
                    torch.empty((B, self.max_annotations), dtype=torch.int64).pin_memory(),
                    torch.empty((B,), dtype=torch.int32).pin_memory())
            self._lane_host = {k: v for k, v in self._lane_host.items() if k[0] != lane}
            self._lane_host[key] = host
        with torch.cuda.stream(stream):
            out, ids, counts = ticket.tensors                     # produced on this stream: no wait needed
            if meta_batch is not None:
                from .annotation import inverse_transform_batch
                out = inverse_transform_batch(out, meta_batch)
            host[0].copy_(out, non_blocking=True)
            host[1].copy_(ids, non_blocking=True)
            host[2].copy_(counts, non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        self.last_nn_time = time.perf_counter() - t0               # host time to queue the network (it runs asynchronously)
        pending = CifCaf.Pending(self, lane, done, host, B, t0, fields=(cif, caf, meta_batch))
        self._lane_pending[lane] = pending
        return pending

    #: multi-GPU (reference ``predictor.py:33-37`` wraps the model in ``nn.DataParallel``: one process, the field
    #: tensors of all GPUs gathered to GPU 0 and copied to the host).  Here: one process per GPU under ``torch.distributed``
    #: (backend ``nccl`` = RCCL over xGMI); every rank is handed the SAME batch, runs network + decode on its shard of the
    #: images and ONE ``all_gather`` of the packed annotation blocks gives every rank the whole batch's annotations -- what
    #: ``batch`` returns on every rank.  ``None`` = automatic (an initialised process group of more than one rank), ``False`` =
    #: never (every rank decodes whole batches on its own), a ``ProcessGroup`` = that group.
    distributed = None

    def _dist(self):
        if self.distributed is False:
            return None
        from . import distributed as dist_mod
        group = None if self.distributed in (None, True) else self.distributed
        act = dist_mod.active(group)
        return None if act is None else (act[0], act[1], group)

    def _batch_sharded(self, model, image_batch, device, meta_batch, rank, world, group):
        """``batch`` under ``torch.distributed``: this rank's shard through network + decode, one collective, the whole batch back."""
        from . import distributed as dist_mod
        from .annotation import inverse_transform_batch
        n = len(image_batch)
        lo, hi, per = dist_mod.shard_batch(n, rank, world)
        pick = list(range(lo, hi)) + [min(lo, n - 1)] * (per - (hi - lo))      # (a short shard is padded with a repeated image)
        start_nn = time.perf_counter()
        local = image_batch[torch.as_tensor(pick, dtype=torch.long, device=image_batch.device)]
        with torch.no_grad():
            if device is not None:
                local = local.to(device, non_blocking=True)
            heads = model(local)
        if local.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.last_nn_time = time.perf_counter() - start_nn
        start_decoder = time.perf_counter()
        out, ids, counts = self.decode_heads(heads)
        if (counts.cpu().numpy() & native.COUNT_FAILED).any() and self.cpp_decoder.pool_overflowed():
            # an image of THIS shard ran out of its tile pool: decoded again here, before the collective (the other ranks wait in it)
            for dec in [self.cpp_decoder]:
                dec.use_full_pool()
            out, ids, counts = self.decode_heads(heads)
        if meta_batch is not None:
            out = inverse_transform_batch(out, [meta_batch[i] for i in pick])
        import torch.distributed as tdist
        if tdist.get_backend(group) != 'nccl':       # (gloo: the CPU tests) host tensors through the collective
            out, ids, counts = out.cpu(), ids.cpu(), counts.cpu()
        out, ids, counts = dist_mod.gather_annotations(out, ids, counts, group=group)      # the ONE collective of the batch
        out, ids, counts = dist_mod.merge_shards(out, ids, counts, n, world)
        result = self._annotations_from_host(out.cpu().numpy(), ids.cpu().numpy(), counts.cpu().numpy())   # raises for a failed image of ANY rank
        self.last_decoder_time = time.perf_counter() - start_decoder
        return result

    def batch(self, model, image_batch, *, device=None, gt_anns_batch=None, meta_batch=None):
        """Image batch -> annotations batch, fields never leave the device.  With ``meta_batch`` (the metas of the
        preprocessing, no rotation) the annotations come back in ORIGINAL-image coordinates: the inverse transform
        (reference ``annotation.py:162-200``) runs on the decoded tensor before its one small D2H copy.
        Under ``torch.distributed`` (see :attr:`distributed`) the batch is sharded over the ranks."""
        d = self._dist()
        if d is not None:
            return self._batch_sharded(model, image_batch, device, meta_batch, *d)
        start_nn = time.perf_counter()
        with torch.no_grad():
            if device is not None:
                image_batch = image_batch.to(device, non_blocking=True)
            heads = model(image_batch)
        if image_batch.is_cuda:
            torch.cuda.current_stream().synchronize()      # make last_nn_time mean what the reference's does
        self.last_nn_time = time.perf_counter() - start_nn

        start_decoder = time.perf_counter()
        out, ids, counts = self.decode_heads(heads)
        if meta_batch is not None:
            from .annotation import inverse_transform_batch
            out = inverse_transform_batch(out, meta_batch)
        out, ids, counts = out.cpu().numpy(), ids.cpu().numpy(), counts.cpu().numpy()   # one small D2H
        if (counts & native.COUNT_FAILED).any() and self.cpp_decoder.pool_overflowed():
            out, ids, counts = self._decode_again_with_full_pool(heads[self.cif_metas[0].head_index],
                                                                 heads[self.caf_metas[0].head_index], meta_batch)
        result = self._annotations_from_host(out, ids, counts)
        self.last_decoder_time = time.perf_counter() - start_decoder
        LOG.debug('time: nn = %.1fms, dec = %.1fms', self.last_nn_time * 1e3, self.last_decoder_time * 1e3)
        return result


class CifCafDense(Decoder):
    """Reference ``decoder/cifcaf.py:17-78``: a third head with dense (extra) bones is decoded together
    with the sparse CAF head -- the two CAF field stacks are concatenated along the bone axis and go
    through the same association kernel with the concatenated skeleton (53+ bones for COCO: the LDS
    variant of the growth state).  Like the reference's C++ (``cifcaf.cpp:299-301``, commented out),
    ``dense_coupling`` only selects the decoder; the confidence scales are recorded, not applied."""
    dense_coupling = 0.0

    def __init__(self, cif_meta: headmeta.Cif, caf_meta: headmeta.Caf, dense_caf_meta: headmeta.Caf):
        super().__init__()
        self.cif_meta = cif_meta
        self.caf_meta = caf_meta
        self.dense_caf_meta = dense_caf_meta
        self.priority += cif_meta.n_fields / 1000.0
        self.priority += caf_meta.n_fields / 1000.0
        self.priority += dense_caf_meta.n_fields / 1000.0
        self.dense_caf_meta.decoder_confidence_scales = [self.dense_coupling for _ in self.dense_caf_meta.skeleton]
        concatenated_caf_meta = headmeta.Caf.concatenate([caf_meta, dense_caf_meta])
        self.cifcaf = CifCaf([cif_meta], [concatenated_caf_meta])

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        group = parser.add_argument_group('CifCafDense decoder')
        group.add_argument('--dense-connections', nargs='?', type=float, default=0.0, const=1.0)

    @classmethod
    def configure(cls, args: argparse.Namespace):
        cls.dense_coupling = args.dense_connections

    @classmethod
    def factory(cls, head_metas):
        if len(head_metas) < 3 or not cls.dense_coupling:
            return []
        return [
            CifCafDense(cif_meta, caf_meta, dense_meta)
            for cif_meta, caf_meta, dense_meta in zip(head_metas, head_metas[1:], head_metas[2:])
            if (isinstance(cif_meta, headmeta.Cif) and isinstance(caf_meta, headmeta.Caf)
                and isinstance(dense_meta, headmeta.Caf))
        ]

    def _fields(self, fields, dim):
        return [fields[self.cif_meta.head_index],
                torch.cat([fields[self.caf_meta.head_index], fields[self.dense_caf_meta.head_index]], dim=dim)]

    def __call__(self, fields, initial_annotations=None):
        return self.cifcaf(self._fields(fields, 0))

    def batch(self, model, image_batch, *, device=None, gt_anns_batch=None):
        """Batched: heads ``[B,A,8,H,W]`` are concatenated along the bone axis on the device."""
        inner = self.cifcaf
        wrapped = lambda images: self._fields(model(images), 1)    # noqa: E731
        result = inner.batch(wrapped, image_batch, device=device, gt_anns_batch=gt_anns_batch)
        self.last_nn_time, self.last_decoder_time = inner.last_nn_time, inner.last_decoder_time
        return result


def _nms_keep(boxes, scores, iou_threshold):
    """Greedy IoU non-maximum suppression on (x0,y0,x1,y1) boxes -> kept indices (what
    ``torchvision.ops.nms`` computes; torchvision is not a dependency here)."""
    order = np.argsort(-scores, kind='stable')
    area = np.maximum(0.0, boxes[:, 2] - boxes[:, 0]) * np.maximum(0.0, boxes[:, 3] - boxes[:, 1])
    keep, alive = [], np.ones(len(boxes), dtype=bool)
    for i in order:
        if not alive[i]:
            continue
        keep.append(i)
        xx0 = np.maximum(boxes[i, 0], boxes[:, 0]); yy0 = np.maximum(boxes[i, 1], boxes[:, 1])
        xx1 = np.minimum(boxes[i, 2], boxes[:, 2]); yy1 = np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.maximum(0.0, xx1 - xx0) * np.maximum(0.0, yy1 - yy0)
        iou = inter / np.maximum(area[i] + area - inter, 1e-12)
        alive &= ~(iou > iou_threshold)
        alive[i] = False
    return np.asarray(keep, dtype=np.int64)


class CifDet(Decoder):
    """Detection decoder on the MI355X (reference ``decoder/cifdet.py:16-91``)."""
    iou_threshold = 0.5
    instance_threshold = 0.15
    nms_by_category = True
    suppression = 0.1

    def __init__(self, head_metas: List[headmeta.CifDet]):
        super().__init__()
        self.metas = head_metas
        self.priority = -1.0                      # prefer keypoints over detections
        self.priority += sum(m.n_fields for m in head_metas) / 1000.0
        self.cpp_decoder = native.CifDet()

    @classmethod
    def factory(cls, head_metas):
        return [cls([meta]) for meta in head_metas if isinstance(meta, headmeta.CifDet)]

    def _post(self, categories, scores, boxes):
        """Reference ``cifdet.py:61-91``: IoU NMS (suppressed scores x0.1), instance threshold, xywh."""
        categories, scores, boxes = np.asarray(categories), np.asarray(scores).copy(), np.asarray(boxes).copy()
        if len(scores):
            if self.nms_by_category:              # batched_nms: boxes of different categories never overlap
                offset = categories.astype(np.float64)[:, None] * (boxes.max() + 1.0)
                keep = _nms_keep(boxes + offset, scores, self.iou_threshold)
            else:
                keep = _nms_keep(boxes, scores, self.iou_threshold)
            pre = scores.copy()
            scores *= self.suppression
            scores[keep] = pre[keep]
        mask = scores > self.instance_threshold
        boxes = boxes[mask]
        boxes[:, 2:] -= boxes[:, :2]
        return [AnnotationDet(self.metas[0].categories).set(int(c), float(s), b)
                for c, s, b in zip(categories[mask], scores[mask], boxes)]

    def __call__(self, fields, initial_annotations=None):
        cat, sc, bx = self.cpp_decoder.call(fields[self.metas[0].head_index], self.metas[0].stride)
        return self._post(cat.cpu().numpy(), sc.cpu().numpy(), bx.cpu().numpy())

    def batch(self, model, image_batch, *, device=None, gt_anns_batch=None):
        start_nn = time.perf_counter()
        with torch.no_grad():
            if device is not None:
                image_batch = image_batch.to(device, non_blocking=True)
            heads = model(image_batch)
        if image_batch.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.last_nn_time = time.perf_counter() - start_nn
        start_decoder = time.perf_counter()
        cat, sc, bx, cnt = self.cpp_decoder.call_batch(heads[self.metas[0].head_index], self.metas[0].stride)
        cat, sc, bx, cnt = cat.cpu().numpy(), sc.cpu().numpy(), bx.cpu().numpy(), cnt.cpu().numpy()
        result = [self._post(cat[b, :cnt[b]], sc[b, :cnt[b]], bx[b, :cnt[b]]) for b in range(len(cnt))]
        self.last_decoder_time = time.perf_counter() - start_decoder
        return result


class Multi(Decoder):
    """Reference ``decoder/multi.py:11-35``: run several decoders on the same fields."""

    def __init__(self, decoders):
        super().__init__()
        self.decoders = decoders

    def __call__(self, all_fields, *, initial_annotations=None):
        out = []
        for task_i, decoder in enumerate(self.decoders):
            if decoder is None:
                out.append(None)
                continue
            out += decoder(all_fields)
        return out

    def batch(self, model, image_batch, *, device=None, gt_anns_batch=None, meta_batch=None):
        if len(self.decoders) == 1:
            kw = {'meta_batch': meta_batch} if meta_batch is not None else {}
            res = self.decoders[0].batch(model, image_batch, device=device, gt_anns_batch=gt_anns_batch, **kw)
            self.last_nn_time = self.decoders[0].last_nn_time
            self.last_decoder_time = self.decoders[0].last_decoder_time
            return res
        return super().batch(model, image_batch, device=device, gt_anns_batch=gt_anns_batch)

    @property
    def supports_device_inverse(self):
        if getattr(self, '_device_inverse_override', None) is not None:
            return self._device_inverse_override
        return len(self.decoders) == 1 and getattr(self.decoders[0], 'supports_device_inverse', False)

    @supports_device_inverse.setter
    def supports_device_inverse(self, value):
        self._device_inverse_override = bool(value)

    @property
    def pipeline_depth(self):
        """Batches that may be in flight through :meth:`batch_async` (0: the decoder has no asynchronous path -- or the batch is
        sharded over the ranks of a ``torch.distributed`` job: ``batch`` then ends in the job's one collective per batch, which every
        rank has to enter in the same order)."""
        if len(self.decoders) == 1 and getattr(self.decoders[0], '_dist', lambda: None)() is not None:
            return 0
        if len(self.decoders) == 1 and hasattr(self.decoders[0], 'batch_async'):
            return max(1, int(self.decoders[0].decoder_workers or 1))
        return 0

    def batch_async(self, model, image_batch, *, device=None, meta_batch=None):
        inner = self.decoders[0]
        pending = inner.batch_async(model, image_batch, device=device, meta_batch=meta_batch)
        collect = pending.result

        def result():                            # the times of the batch that was collected last, like batch()
            out = collect()
            self.last_nn_time, self.last_decoder_time = inner.last_nn_time, inner.last_decoder_time
            return out
        pending.result = result
        return pending


DECODERS = {CifCaf, CifCafDense, CifDet}        # + tracking.TrackingPose, which registers itself on import


def _with_tracking():
    """``DECODERS`` including the tracking decoders (their module imports this one, hence the late import)."""
    from . import tracking      # noqa: F401
    return DECODERS


def cli(parser, *, workers=None):
    """Reference ``decoder/factory.py:20-49``."""
    group = parser.add_argument_group('decoder configuration')
    available = [dec.__name__.lower() for dec in _with_tracking()]
    group.add_argument('--decoder', default=None, nargs='+',
                       help='Decoders to be considered: {}.'.format(available))
    group.add_argument('--seed-threshold', default=native.CifSeeds.get_threshold(), type=float,
                       help='minimum threshold for seeds')
    group.add_argument('--instance-threshold', type=float, default=None,
                       help='filter instances by score (default is 0.0 with --force-complete-pose '
                            'and {} otherwise)'.format(native.NMSKeypoints.get_instance_threshold()))
    group.add_argument('--decoder-workers', default=workers, type=int,
                       help='batches decoded at once: decode lanes (stream + workspace each) beside the next '
                            'batch\'s network; default {}'.format(CifCaf.decoder_workers))
    group = parser.add_argument_group('CifCaf decoders')
    group.add_argument('--cif-th', default=native.CifHr.get_threshold(), type=float, help='cif threshold')
    group.add_argument('--caf-th', default=native.CafScored.get_default_score_th(), type=float,
                       help='caf threshold')
    for dec in _with_tracking():
        dec.cli(parser)
    from . import tracking
    tracking.TrackBase.cli(parser)              # reference decoder/factory.py:47


def configure(args):
    """Reference ``decoder/factory.py:52-82``."""
    if args.instance_threshold is None:
        args.instance_threshold = 0.0 if args.force_complete_pose else native.NMSKeypoints.get_instance_threshold()
    Factory.decoder_request_from_args(args.decoder)
    native.CifHr.set_threshold(args.cif_th)
    native.CifSeeds.set_threshold(args.seed_threshold)
    native.CafScored.set_default_score_th(args.caf_th)
    native.NMSKeypoints.set_instance_threshold(args.instance_threshold)
    CifDet.instance_threshold = args.instance_threshold
    if getattr(args, 'decoder_workers', None) is not None:      # reference decoder/factory.py:66-71 sizes its fork pool here
        CifCaf.decoder_workers = max(1, int(args.decoder_workers))
    for dec in _with_tracking():
        dec.configure(args)
    from . import tracking
    tracking.TrackBase.configure(args)          # reference decoder/factory.py:80


class Factory:
    """Reference ``decoder/factory.py:85-160``."""
    decoder_request: Optional[dict] = None

    @classmethod
    def decoder_request_from_args(cls, list_str):
        if list_str is None:
            cls.decoder_request = None
            return
        cls.decoder_request = defaultdict(list)
        for dec_str in list_str:
            if ':' not in dec_str:
                if dec_str not in cls.decoder_request:
                    cls.decoder_request[dec_str] = []
                continue
            dec_str, _, index = dec_str.partition(':')
            cls.decoder_request[dec_str].append(int(index))

    @classmethod
    def decoders(cls, head_metas):
        def per_class(request, dec_class):
            class_name = dec_class.__name__.lower()
            if request is not None and class_name not in request:
                return []
            decoders = sorted(dec_class.factory(head_metas), key=lambda d: d.priority, reverse=True)
            for dec_i, dec in enumerate(decoders):
                dec.request_index = dec_i
            if request is not None:
                indices = set(request[class_name])
                decoders = (d for i, d in enumerate(decoders) if i in indices)
            return decoders

        decoders = [d for dec_class in _with_tracking() for d in per_class(cls.decoder_request, dec_class)]
        decoders = list(sorted(decoders, key=lambda d: d.priority, reverse=True))
        if not decoders:
            LOG.warning('no decoders found for heads %s', [meta.name for meta in head_metas])
        elif len(decoders) > 1 and cls.decoder_request is None:
            decoders = [decoders[0]]
        return decoders

    @classmethod
    def __call__(cls, head_metas):
        return Multi(cls.decoders(head_metas))


factory = Factory.__call__

