"""``Predictor``: the reference's convenience entry point (reference ``predictor.py:12-192``)
over the device-resident decode path.

Same constructor/attribute/method surface for the inputs that work offline
(``numpy_image(s)``, ``pil_image(s)``, ``image(s)``, ``tensor_batch``); the model
is a randomly initialised ``Shell`` unless one is passed (no checkpoints offline).
Preprocessing restates the evaluation pipeline without torchvision:
``RescaleAbsolute(long_edge, fast=Predictor.fast_rescaling)`` (reference ``transforms/scale.py:42-63,150-174``;
``predictor.py:17,88``).  Like the reference's Predictor the DEFAULT is ``fast_rescaling = True``: without OpenCV
(absent in this image) that is Pillow's antialiased ``BILINEAR`` resize (``scale.py:55-58``) -- on the host Pillow
itself, on the device :func:`resize_bilinear_u8`, a restatement of Pillow's 8-bit resampler that is pixel-equal to
it; ``--precise-rescaling`` selects ``scipy.ndimage.zoom(order=1)`` (``scale.py:59-67``), restated for host and
device by :func:`zoom_linear_u8`.  Then ``CenterPad(long_edge)`` for batch > 1 / ``CenterPadTight(16)`` for batch 1
(reference ``transforms/pad.py:15-112``) and the ImageNet normalisation (reference ``transforms/__init__.py:26-33``).
"""
import argparse
import logging
import math

import numpy as np
import torch

from . import decoder, network

LOG = logging.getLogger(__name__)

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _to_pil(image):
    import PIL.Image
    if isinstance(image, PIL.Image.Image):
        return image.convert('RGB')
    return PIL.Image.fromarray(np.asarray(image, dtype=np.uint8)).convert('RGB')


PIL_PRECISION_BITS = 32 - 8 - 2      # Pillow's fixed-point coefficient precision for 8-bit images


def _pil_bilinear_coeffs(in_size, out_size):
    """Pillow's coefficient table for the BILINEAR (triangle) filter along one axis, as its 8-bit resampler uses
    it: per output sample the first input sample and ``ksize`` fixed-point weights.  The filter support grows with
    the downscale factor (antialiasing); weights are normalised in double, in Pillow's summation order, then
    rounded half away from zero to 22 fractional bits.  Computed on the host (a few hundred values)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    inv = 1.0 / filterscale
    first = np.trunc(center - support + 0.5).astype(np.int64)
    first[first < 0] = 0
    count = np.trunc(center + support + 0.5).astype(np.int64)
    count[count > in_size] = in_size
    count -= first
    k = np.zeros((out_size, ksize), dtype=np.float64)
    total = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):                            # sequential accumulation like the C loop
        arg = np.abs((x + first - center + 0.5) * inv)
        w = np.where((arg < 1.0) & (x < count), 1.0 - arg, 0.0)
        k[:, x] = w
        total = total + w
    nz = total != 0.0
    k[nz] = k[nz] / total[nz, None]
    one = float(1 << PIL_PRECISION_BITS)
    fixed = np.where(k < 0, np.trunc(-0.5 + k * one), np.trunc(0.5 + k * one)).astype(np.int64)
    return first, fixed


def _pil_resample_axis(frame, axis, out_size):
    in_size = frame.shape[axis]
    first, fixed = _pil_bilinear_coeffs(in_size, out_size)
    dev = frame.device
    first_t, fixed_t = torch.from_numpy(first).to(dev), torch.from_numpy(fixed).to(dev)
    shape = [frame.shape[0], frame.shape[1], frame.shape[2]]
    shape[axis] = out_size
    wshape = [1, 1, 1]
    wshape[axis] = out_size
    acc = torch.full(shape, 1 << (PIL_PRECISION_BITS - 1), dtype=torch.int64, device=dev)
    for x in range(fixed.shape[1]):                   # taps beyond a sample's count carry weight 0
        idx = (first_t + x).clamp_(max=in_size - 1)
        acc += frame.index_select(axis, idx).to(torch.int64) * fixed_t[:, x].view(wshape)
    return (acc >> PIL_PRECISION_BITS).clamp_(0, 255).to(torch.uint8)


def resize_bilinear_u8(frame, target_h, target_w):
    """The reference Predictor's DEFAULT rescale where OpenCV is not installed: ``image.resize((tw, th), BILINEAR)``
    of Pillow (reference ``transforms/scale.py:55-58`` with ``predictor.py:17,88``), restated with torch integer ops
    so that it runs on the device: a separable triangle filter whose support scales with the downscale factor,
    22-bit fixed-point weights, the horizontal pass rounded to uint8 before the vertical one -- Pillow's 8-bit
    resampler.  Pixel-EQUAL to Pillow (``tests/test_abi_and_host.py``).
    ``frame``: uint8 ``[H, W, C]`` tensor on any device -> uint8 ``[target_h, target_w, C]``."""
    h, w = frame.shape[:2]
    if target_w != w:
        frame = _pil_resample_axis(frame, 1, target_w)
    if target_h != h:
        frame = _pil_resample_axis(frame, 0, target_h)
    return frame


def zoom_linear_u8(frame, target_h, target_w):
    """The reference's ``--precise-rescaling`` rescale, ``scipy.ndimage.zoom(im, (th / h, tw / w, 1), order=1)`` on a
    uint8 image (reference ``transforms/scale.py:59-67``), restated with torch ops so that it runs on the device: corner-aligned
    coordinates ``j * (n_in - 1) / (n_out - 1)``, linear weights, the sum formed in double in scipy's order, rounded
    like its uint8 output (``(uint8)(t + 0.5)``).  Pixel-equal to scipy (``tests/test_abi_and_host.py``).
    ``frame``: uint8 ``[H, W, C]`` tensor on any device -> uint8 ``[target_h, target_w, C]``."""
    h, w = frame.shape[:2]
    dev = frame.device

    def axis(n_in, n_out):
        zoom = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        cc = torch.arange(n_out, dtype=torch.float64, device=dev) * zoom
        start = torch.floor(cc)
        w0 = 1.0 - (cc - start)                      # each weight is 1 - |distance| of its own sample
        w1 = 1.0 - ((start + 1.0) - cc)
        i0 = start.to(torch.int64).clamp_(0, n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)           # its weight is 0 wherever the clamp acts
        outside = cc > (n_in - 1)                    # rounding can push the last coordinate past the edge: scipy's
        return i0, i1, w0, w1, outside               # 'constant' mode then yields cval = 0 for that row / column
    y0, y1, wy0, wy1, oy = axis(h, target_h)
    x0, x1, wx0, wx1, ox = axis(w, target_w)
    f = frame.to(torch.float64)
    wy0, wy1 = wy0.view(-1, 1, 1), wy1.view(-1, 1, 1)
    wx0, wx1 = wx0.view(1, -1, 1), wx1.view(1, -1, 1)
    t = (f[y0][:, x0] * wy0) * wx0                   # (value * w_y) * w_x, summed in scipy's order
    t = t + (f[y0][:, x1] * wy0) * wx1
    t = t + (f[y1][:, x0] * wy1) * wx0
    t = t + (f[y1][:, x1] * wy1) * wx1
    t = torch.where(oy.view(-1, 1, 1) | ox.view(1, -1, 1), torch.zeros_like(t), t)
    return torch.floor(t.clamp_(0.0, 255.0) + 0.5).to(torch.uint8)


def _target_size(w0, h0, long_edge):
    s = long_edge / max(h0, w0)                       # RescaleAbsolute, transforms/scale.py:160-165
    return (int(w0 * s), int(long_edge)) if h0 > w0 else (int(long_edge), int(h0 * s))


def preprocess_image(image, *, long_edge=None, batch_mode=False, fast=True):
    """-> (float32 tensor [3,H,W], meta) with the meta fields ``inverse_transform`` needs.  ``fast`` (the
    reference Predictor's default, ``predictor.py:17``): Pillow's antialiased bilinear resize; ``fast=False``
    (``--precise-rescaling``): scipy's order-1 zoom."""
    import PIL.Image
    image = _to_pil(image)
    w0, h0 = image.size
    meta = {'offset': np.array((0.0, 0.0)), 'scale': np.array((1.0, 1.0)), 'hflip': False,
            'rotation': {'angle': 0.0, 'width': None, 'height': None},
            'valid_area': np.array((0.0, 0.0, w0 - 1, h0 - 1)), 'width_height': np.array((w0, h0))}
    if long_edge:
        tw, th = _target_size(w0, h0, long_edge)
        if fast:
            image = image.resize((tw, th), getattr(PIL.Image, 'Resampling', PIL.Image).BILINEAR)
        else:                                         # --precise-rescaling: scipy.ndimage.zoom, order 1
            image = PIL.Image.fromarray(zoom_linear_u8(torch.from_numpy(np.array(image)), th, tw).numpy())
        sx, sy = (tw - 1) / (w0 - 1), (th - 1) / (h0 - 1)
        meta['offset'] *= (sx, sy)
        meta['scale'] *= (sx, sy)
        meta['valid_area'][:2] *= (sx, sy)
        meta['valid_area'][2:] *= (sx, sy)
    w, h = image.size
    if batch_mode:
        assert long_edge, '--long-edge must be provided for batch size > 1'
        tw, th, fill = long_edge, long_edge, (124, 116, 104)
    else:
        tw = math.ceil((w - 1) / 16) * 16 + 1
        th = math.ceil((h - 1) / 16) * 16 + 1
        fill = (124, 116, 104)
    left, top = max(0, int((tw - w) / 2.0)), max(0, int((th - h) / 2.0))
    canvas = PIL.Image.new('RGB', (max(tw, w), max(th, h)), fill)
    canvas.paste(image, (left, top))
    meta['offset'] -= (left, top)
    meta['valid_area'][:2] += (left, top)
    x = np.asarray(canvas, dtype=np.float32) / 255.0
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))), meta


def _as_u8_rgb(image):
    """-> contiguous uint8 ``[H, W, 3]`` numpy array (what ``_to_pil(image)`` holds)."""
    if isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3:
        return np.ascontiguousarray(image)
    return np.ascontiguousarray(np.asarray(_to_pil(image), dtype=np.uint8))


class _Staging:
    """Two pinned host buffers per (device, shape) for the frames of a batch: frames are copied in with memcpy and go to the
    device in ONE asynchronous upload.  (Pinning every frame on its own costs a pinned allocation per frame -- milliseconds
    each, and the runtime may wait for the device -- and an upload from pageable memory makes the host wait for the stream.)
    A buffer is reused only after the upload that read it last has finished (its event)."""
    _slots = {}

    @classmethod
    def get(cls, device, shape):
        key = (str(device), tuple(shape))
        ring = cls._slots.get(key)
        if ring is None:
            if len(cls._slots) > 8:                    # a stream of differently sized batches: do not hoard pinned memory
                cls._slots.clear()
            ring = cls._slots[key] = {'bufs': [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)],
                                      'events': [None, None], 'next': 0}
        k = ring['next']
        ring['next'] = 1 - k
        if ring['events'][k] is not None:
            ring['events'][k].synchronize()
        return ring, k


def preprocess_batch_device(images, *, long_edge, device, fast=True):
    """Device-side form of ``preprocess_image`` for batch mode: the uint8 frames are
    uploaded as they are (a quarter of the bytes of normalised float32), rescaled to ``long_edge`` with the
    reference's own arithmetic (:func:`resize_bilinear_u8` = Pillow, or with ``fast=False``
    :func:`zoom_linear_u8` = scipy; rounded to uint8 like the host path), centre-padded to
    ``long_edge x long_edge`` with the reference's fill colour and normalised -- all on ``device``.
    -> (float32 ``[B,3,long_edge,long_edge]`` on ``device``, metas).  The pixels EQUAL the host path's."""
    assert long_edge, '--long-edge must be provided for batch size > 1'
    mean, std, fill, d255 = _device_constants(device)
    canvas = fill.expand(len(images), long_edge, long_edge, 3).contiguous()
    frames = [_as_u8_rgb(image) for image in images]
    on_gpu = torch.device(device).type == 'cuda'
    same = on_gpu and len({f.shape for f in frames}) == 1
    stack = None
    if same:                                          # the usual case (a video, a batch of one camera): one pinned upload
        ring, k = _Staging.get(device, (len(frames),) + frames[0].shape)
        host = ring['bufs'][k].numpy()
        for b, f in enumerate(frames):
            host[b] = f
        stack = ring['bufs'][k].to(device, non_blocking=True)
        ring['events'][k] = torch.cuda.Event()
        ring['events'][k].record(torch.cuda.current_stream(device))
    metas = []
    for b, frame in enumerate(frames):
        h0, w0 = frame.shape[:2]
        tw, th = _target_size(w0, h0, long_edge)
        x = stack[b] if stack is not None else torch.from_numpy(frame).to(device)
        if (th, tw) != (h0, w0):
            x = (resize_bilinear_u8 if fast else zoom_linear_u8)(x, th, tw)
        left, top = max(0, int((long_edge - tw) / 2.0)), max(0, int((long_edge - th) / 2.0))
        if stack is not None and (th, tw) == (h0, w0):
            if b == 0:                                # all frames alike and no rescale: one placement for the batch
                canvas[:, top:top + th, left:left + tw] = stack
        else:
            canvas[b, top:top + th, left:left + tw] = x
        sx, sy = (tw - 1) / (w0 - 1), (th - 1) / (h0 - 1)
        metas.append({'offset': np.array((-float(left), -float(top))), 'scale': np.array((sx, sy)), 'hflip': False,
                      'rotation': {'angle': 0.0, 'width': None, 'height': None},
                      'valid_area': np.array((left, top, (w0 - 1) * sx, (h0 - 1) * sy), dtype=np.float64),
                      'width_height': np.array((w0, h0))})
    # ToTensor + Normalize exactly as the host path does them: float32(u8) / 255, then (x - mean) / std
    # (a DEVICE tensor as divisor: dividing by a Python scalar is turned into a multiplication by 1/255 on the GPU,
    # which rounds differently from the host's true division)
    batch = canvas.permute(0, 3, 1, 2).to(torch.float32) / d255
    return ((batch - mean) / std).contiguous(), metas


_constants = {}


def _device_constants(device):
    """ImageNet mean / std, the padding colour and 255.0 as tensors on ``device`` (uploaded once: every ``torch.tensor(...,
    device=...)`` is a blocking copy)."""
    key = str(device)
    if key not in _constants:
        _constants[key] = (torch.tensor(IMAGENET_MEAN, device=device, dtype=torch.float32).view(3, 1, 1),
                           torch.tensor(IMAGENET_STD, device=device, dtype=torch.float32).view(3, 1, 1),
                           torch.tensor((124, 116, 104), dtype=torch.uint8, device=device).view(1, 1, 1, 3),
                           torch.full((1,), 255.0, dtype=torch.float32, device=device))
    return _constants[key]


class Predictor:
    """Predict from various inputs with a common configuration."""
    device_preprocess = False      #: batch mode: rescale / pad / normalise on the device instead of with PIL
    pipelined = True               #: overlap batch i's decode with batch i+1's preprocessing + network (decoder lanes)
    batch_size = 1
    device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
    fast_rescaling = True          #: reference predictor.py:17 (False = --precise-rescaling, scipy zoom)
    long_edge = None
    base_name = 'resnet50'
    channels_last = True
    dtype = torch.float32          #: backbone compute dtype (heads always emit float32 fields)
    conv1x1_choices = None         #: a ``fused.choices()`` table to adopt (multi-rank jobs: rank 0's, so all ranks run the same kernels)

    def __init__(self, checkpoint=None, head_metas=None, *, model=None, json_data=False):
        if checkpoint is not None and model is None:
            self.base_name = str(checkpoint)
        self.json_data = json_data
        if self.conv1x1_choices is not None:
            from . import fused
            fused.set_choices(self.conv1x1_choices, replace=True)
        self.model_cpu = model if model is not None else network.factory(self.base_name, head_metas)
        self.model = self.model_cpu.to(self.device)
        if self.channels_last and self.device.type == 'cuda':
            self.model = self.model.to(memory_format=torch.channels_last)
        self.processor = decoder.factory(self.model_cpu.head_metas)
        self.last_decoder_time = 0.0
        self.last_nn_time = 0.0
        self.total_nn_time = 0.0
        self.total_decoder_time = 0.0
        self.total_images = 0

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser, *, skip_batch_size=False, skip_loader_workers=False):
        group = parser.add_argument_group('Predictor')
        if not skip_batch_size:
            group.add_argument('--batch-size', default=cls.batch_size, type=int, help='processing batch size')
        group.add_argument('--long-edge', default=cls.long_edge, type=int,
                           help='rescale the long side of the image (aspect ratio maintained)')
        group.add_argument('--precise-rescaling', dest='fast_rescaling', default=True, action='store_false',
                           help='use more exact image rescaling (requires scipy)')      # reference predictor.py:72-74
        group.add_argument('--basenet', default=cls.base_name, choices=sorted(network.BASE_FACTORIES))

    @classmethod
    def configure(cls, args: argparse.Namespace):
        cls.batch_size = args.batch_size
        cls.long_edge = args.long_edge
        cls.fast_rescaling = getattr(args, 'fast_rescaling', cls.fast_rescaling)
        cls.base_name = getattr(args, 'basenet', cls.base_name)
        if getattr(args, 'device', None) is not None:
            cls.device = args.device

    def _forward(self, image_batch):
        image_batch = image_batch.to(self.device, non_blocking=True)
        if self.channels_last and image_batch.is_cuda:
            image_batch = image_batch.contiguous(memory_format=torch.channels_last)
        if self.dtype != torch.float32 and image_batch.is_cuda:
            with torch.autocast('cuda', dtype=self.dtype):
                return self.model(image_batch)
        return self.model(image_batch)

    def _device_inverse(self, meta_batch):
        """Can pad / rescale / flip be undone on the decoded tensor, before its one D2H copy?"""
        return (meta_batch is not None and getattr(self.processor, 'supports_device_inverse', False)
                and self.device.type == 'cuda'
                and all((m.get('rotation') or {}).get('angle', 0.0) == 0.0 and not m.get('horizontal_swap')
                        for m in meta_batch))

    def _finish(self, pred_batch, meta_batch, n_images):
        self.last_decoder_time = self.processor.last_decoder_time
        self.last_nn_time = self.processor.last_nn_time
        self.total_decoder_time += self.last_decoder_time
        self.total_nn_time += self.last_nn_time
        self.total_images += n_images
        if meta_batch is None:
            meta_batch = [None] * len(pred_batch)
        out = []
        for pred, meta in zip(pred_batch, meta_batch):
            if meta is not None:                      # (None: pad / rescale / flip were undone on the device already)
                pred = [ann.inverse_transform(meta) for ann in pred]
            if self.json_data:
                pred = [ann.json_data() for ann in pred]
            out.append(pred)
        return out

    def tensor_batch(self, processed_image_batch, meta_batch=None):
        """Predict from an already preprocessed ``[B,3,H,W]`` batch -> list (per image) of predictions."""
        model = _CallModel(self._forward)
        if self._device_inverse(meta_batch):   # pad / rescale / flip undone on the decoded tensor, before its one D2H copy
            pred_batch = self.processor.batch(model, processed_image_batch, device=None, meta_batch=meta_batch)
            meta_batch = [None] * len(pred_batch)
        else:
            pred_batch = self.processor.batch(model, processed_image_batch, device=None)
        return self._finish(pred_batch, meta_batch, len(processed_image_batch))

    def tensor_batch_async(self, processed_image_batch, meta_batch=None):
        """``tensor_batch`` in two halves: queue network + decode now, collect later -> a callable returning what
        ``tensor_batch`` returns.  Call it after submitting the NEXT batch and the two overlap (the reference overlaps
        them with ``--decoder-workers`` CPU processes, decoder/decoder.py:33-47)."""
        model = _CallModel(self._forward)
        n = len(processed_image_batch)
        if self._device_inverse(meta_batch):
            pending = self.processor.batch_async(model, processed_image_batch, device=None, meta_batch=meta_batch)
            meta_batch = [None] * n
        else:
            pending = self.processor.batch_async(model, processed_image_batch, device=None)
        return lambda: self._finish(pending.result(), meta_batch, n)

    def _preprocess(self, images):
        batch_mode = self.batch_size > 1
        if batch_mode and self.device_preprocess and self.device.type == 'cuda':
            # on a stream of its own: uploads and resizes of batch i+1 do not queue behind the network of batch i (nor does
            # the host behind them); the network's stream waits for the finished batch only
            if getattr(self, '_pre_stream', None) is None:
                self._pre_stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self._pre_stream):
                batch, metas = preprocess_batch_device(images, long_edge=self.long_edge, device=self.device,
                                                       fast=self.fast_rescaling)
                ready = torch.cuda.Event()
                ready.record(self._pre_stream)
            main.wait_event(ready)
            batch.record_stream(main)
            return batch, metas
        items = [preprocess_image(im, long_edge=self.long_edge, batch_mode=batch_mode, fast=self.fast_rescaling)
                 for im in images]
        return torch.stack([t for t, _ in items]), [m for _, m in items]

    def _images(self, images):
        """Batches of ``batch_size`` images -> (pred, gt, meta) per image, in order.  With an asynchronous decoder
        (``CifCaf.batch_async``) the batches are pipelined: batch i+1 is preprocessed and queued while batch i decodes;
        at most ``pipeline_depth`` (= ``--decoder-workers``) batches are in flight."""
        depth = getattr(self.processor, 'pipeline_depth', 0) if self.device.type == 'cuda' and self.pipelined else 0
        if depth < 1:
            for i in range(0, len(images), self.batch_size):
                batch, metas = self._preprocess(images[i:i + self.batch_size])
                for pred, meta in zip(self.tensor_batch(batch, metas), metas):
                    yield pred, [], meta
            return
        in_flight = []                                  # (collect, metas), oldest first
        for i in range(0, len(images), self.batch_size):
            if len(in_flight) >= depth:                 # every lane is taken: the oldest batch's result first
                collect, m = in_flight.pop(0)
                for pred, meta in zip(collect(), m):
                    yield pred, [], meta
            batch, metas = self._preprocess(images[i:i + self.batch_size])
            in_flight.append((self.tensor_batch_async(batch, metas), metas))
        for collect, m in in_flight:
            for pred, meta in zip(collect(), m):
                yield pred, [], meta

    def numpy_images(self, numpy_images):
        yield from self._images(list(numpy_images))

    def numpy_image(self, image):
        return next(iter(self.numpy_images([image])))

    def pil_images(self, pil_images):
        yield from self._images(list(pil_images))

    def pil_image(self, image):
        return next(iter(self.pil_images([image])))

    def images(self, file_names):
        import PIL.Image
        yield from self._images([PIL.Image.open(f).convert('RGB') for f in file_names])

    def image(self, file_name):
        return next(iter(self.images([file_name])))


class _CallModel:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)
