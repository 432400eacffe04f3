"""Pose annotation object the decoder returns: the part of the reference's
``openpifpaf.annotation.Annotation`` (reference ``annotation.py:16-143``) that
``Predictor`` and JSON output use."""
import numpy as np


class Annotation:
    def __init__(self, keypoints, skeleton, sigmas=None, *, categories=None, score_weights=None):
        self.keypoints = keypoints
        self.skeleton = skeleton
        self.sigmas = sigmas
        self.categories = categories
        self.score_weights = score_weights
        self.category_id = 1
        self.data = np.zeros((len(keypoints), 3), dtype=np.float32)          # x, y, v
        self.joint_scales = np.zeros((len(keypoints),), dtype=np.float32)
        self.fixed_score = None
        self.fixed_bbox = None
        self.decoding_order = []
        self.frontier_order = []
        if score_weights is None:
            self.score_weights = np.ones((len(keypoints),))
        else:
            assert len(score_weights) == len(keypoints), 'wrong number of scores'
            self.score_weights = np.asarray(score_weights, dtype=np.float64)
        self.score_weights = self.score_weights / np.sum(self.score_weights)

    @property
    def score(self):
        """Weighted instance score over the confidence-sorted joints (reference
        ``annotation.py:98-110``); NOT the plain mean the native NMS uses."""
        if self.fixed_score is not None:
            return self.fixed_score
        v = self.data[:, 2]
        return float(np.sum(self.score_weights * np.sort(v)[::-1]))

    def bbox(self):
        if self.fixed_bbox is not None:
            return self.fixed_bbox
        m = self.data[:, 2] > 0
        if not np.any(m):
            return [0, 0, 0, 0]
        x = self.data[m, 0] - self.joint_scales[m]
        y = self.data[m, 1] - self.joint_scales[m]
        w = self.data[m, 0] + self.joint_scales[m]
        h = self.data[m, 1] + self.joint_scales[m]
        x0, y0 = float(x.min()), float(y.min())
        return [x0, y0, float(w.max()) - x0, float(h.max()) - y0]

    def json_data(self, coordinate_digits=2):
        """Reference ``annotation.py:121-143``."""
        v_mask = self.data[:, 2] > 0.0
        keypoints = np.copy(self.data)
        keypoints[v_mask, 2] = np.maximum(0.01, keypoints[v_mask, 2])
        keypoints = np.around(keypoints.astype(np.float64), coordinate_digits)
        data = {
            'keypoints': keypoints.reshape(-1).tolist(),
            'bbox': [round(float(c), coordinate_digits) for c in self.bbox()],
            'score': max(0.001, round(self.score, 3)),
            'category_id': self.category_id,
        }
        id_ = getattr(self, 'id_', None)
        if id_:
            data['id_'] = id_
        return data

    def inverse_transform(self, meta):
        """Undo pad / rescale / hflip recorded in ``meta`` (reference ``annotation.py:162-200``)."""
        import copy
        import math
        ann = copy.deepcopy(self)
        if meta is None:
            return ann
        rot = meta.get('rotation')
        if rot is not None and rot.get('angle', 0.0) != 0.0:
            angle = -rot['angle']
            rw, rh = rot['width'], rot['height']
            ca, sa = math.cos(angle / 180.0 * math.pi), math.sin(angle / 180.0 * math.pi)
            x_old = ann.data[:, 0].copy() - (rw - 1) / 2
            y_old = ann.data[:, 1].copy() - (rh - 1) / 2
            ann.data[:, 0] = (rw - 1) / 2 + ca * x_old + sa * y_old
            ann.data[:, 1] = (rh - 1) / 2 - sa * x_old + ca * y_old
        offset = np.asarray(meta.get('offset', (0.0, 0.0)), dtype=np.float32)
        scale = np.asarray(meta.get('scale', (1.0, 1.0)), dtype=np.float32)
        ann.data[:, 0] += offset[0]
        ann.data[:, 1] += offset[1]
        ann.data[:, 0] = ann.data[:, 0] / scale[0]
        ann.data[:, 1] = ann.data[:, 1] / scale[1]
        ann.joint_scales = ann.joint_scales / scale[0]
        if meta.get('hflip'):
            w = meta['width_height'][0]
            ann.data[:, 0] = -ann.data[:, 0] + (w - 1)
            if meta.get('horizontal_swap'):
                ann.data[:] = meta['horizontal_swap'](ann.data)
        return ann


_meta_staging = {}     # pinned upload buffers of inverse_transform_batch (a small ring per device and batch size)


def inverse_transform_batch(annotations, metas):
    """``Annotation.inverse_transform`` (reference ``annotation.py:162-200``) for a whole decoded batch at
    once, on the device the decoder left it on: ``annotations`` ``[B, max, K, 4]`` (v, x, y, s) as returned by
    ``native.CifCaf.call_batch`` -> same shape in original-image coordinates.  Offsets, scales and the
    horizontal flip are handled; rotated inputs (a training-time augmentation) are not."""
    import torch
    B = annotations.shape[0]
    assert len(metas) == B
    rows = []
    for m in metas:
        m = m or {}
        rot = m.get('rotation')
        if rot is not None and rot.get('angle', 0.0) != 0.0:
            raise NotImplementedError('inverse_transform_batch: rotated inputs')
        off, sc = m.get('offset', (0.0, 0.0)), m.get('scale', (1.0, 1.0))
        flip = bool(m.get('hflip'))
        rows.append([float(off[0]), float(off[1]), float(sc[0]), float(sc[1]), 1.0 if flip else 0.0,
                     float(m['width_height'][0]) - 1.0 if flip else 0.0])
    t = torch.tensor(rows, dtype=annotations.dtype)
    if annotations.is_cuda:                 # through pinned memory: a pageable upload would make the host wait for the stream
        ring = _meta_staging.setdefault((str(annotations.device), B, annotations.dtype), {'bufs': [torch.empty((B, 6), dtype=annotations.dtype).pin_memory() for _ in range(4)], 'events': [None] * 4, 'next': 0})
        k = ring['next']
        ring['next'] = (k + 1) % 4
        if ring['events'][k] is not None:
            ring['events'][k].synchronize()
        ring['bufs'][k].copy_(t)
        t = ring['bufs'][k].to(annotations.device, non_blocking=True)
        ring['events'][k] = torch.cuda.Event()
        ring['events'][k].record(torch.cuda.current_stream(annotations.device))
    t = t.view(B, 1, 1, 6)
    out = annotations.clone()
    out[..., 1] = (annotations[..., 1] + t[..., 0]) / t[..., 2]
    out[..., 2] = (annotations[..., 2] + t[..., 1]) / t[..., 3]
    out[..., 3] = annotations[..., 3] / t[..., 2]
    out[..., 1] = torch.where(t[..., 4] > 0, -out[..., 1] + t[..., 5], out[..., 1])
    return out


class AnnotationDet:
    """Detection annotation (reference ``annotation.py:216-262``)."""

    def __init__(self, categories):
        self.categories = categories
        self.category_id = None
        self.score = None
        self.bbox = None

    def set(self, category_id, score, bbox):
        self.category_id = category_id
        self.score = score
        self.bbox = np.asarray(bbox)
        return self

    @property
    def category(self):
        return self.categories[self.category_id - 1]

    def json_data(self, coordinate_digits=2):
        return {
            'category_id': self.category_id,
            'category': self.category,
            'score': max(0.001, round(float(self.score), 3)),
            'bbox': [round(float(c), coordinate_digits) for c in self.bbox],
        }

    def inverse_transform(self, meta):
        import copy
        ann = copy.deepcopy(self)
        if meta is None:
            return ann
        ann.bbox = np.asarray(ann.bbox, dtype=np.float64).copy()
        ann.bbox[:2] += np.asarray(meta.get('offset', (0.0, 0.0)))
        ann.bbox[:2] /= np.asarray(meta.get('scale', (1.0, 1.0)))
        ann.bbox[2:] /= np.asarray(meta.get('scale', (1.0, 1.0)))
        if meta.get('hflip'):
            w = meta['width_height'][0]
            ann.bbox[0] = -(ann.bbox[0] + ann.bbox[2]) - 1.0 + w
        return ann
