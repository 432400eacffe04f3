"""Pose tracking on top of the HIP decode path: the reference's ``TrackingPose`` decoder.

The reference tracks by decoding a "tracking pose" -- the current frame's 17 joints plus the previous
frame's 17 joints, connected by 17 temporal bones -- with the ordinary CifCaf decoder, seeded with the
previous frame's poses as initial annotations (``decoder/tracking_pose.py:18-300``,
``decoder/track_base.py:11-168``, ``decoder/track_annotation.py:4-57``).  Here that decode is ONE call of
the association kernel with ``n_keypoints = 34 > 17`` CIF fields (``include/openpifpaf_amd.h``:
``opa_shape.n_keypoints``); everything in this module is the host-side bookkeeping around it: tracks,
their scores, the soft NMS between tracks and the pruning rules.
"""
import argparse
import logging
import time
from typing import List

import numpy as np
import torch

from . import headmeta
from .annotation import Annotation
from . import decoder as _decoder
from . import native
from .decoder import CifCaf, Decoder

LOG = logging.getLogger(__name__)


class Occupancy:
    """Host-side occupancy map with the semantics of the reference's C++ class (``csrc/src/occupancy.cpp``
    ``:13-79``) for the few hundred box tests of the soft NMS between tracks; the decode itself never
    leaves the device."""

    def __init__(self, reduction=2.0, min_scale=4.0):
        self.reduction = float(reduction)
        self.min_scale_reduced = float(min_scale) / float(reduction)
        self.occupancy = np.zeros((0, 1, 1), dtype=np.uint8)

    def reset(self, shape):                                      # occupancy.cpp:46-68
        n, h, w = shape
        self.occupancy = np.zeros((int(n), int(h / self.reduction) + 1, int(w / self.reduction) + 1), dtype=np.uint8)

    def clear(self):
        self.occupancy[:] = 0

    def n_fields(self):
        return self.occupancy.shape[0]

    def set(self, f, x, y, sigma):                               # occupancy.cpp:13-29
        x, y, sigma = float(x), float(y), float(sigma)
        if self.reduction != 1.0:
            x /= self.reduction
            y /= self.reduction
            sigma = max(self.min_scale_reduced, sigma / self.reduction)
        _, h, w = self.occupancy.shape
        minx = min(max(int(x - sigma), 0), w - 1)
        miny = min(max(int(y - sigma), 0), h - 1)
        maxx = min(max(int(x + sigma), minx + 1), w)
        maxy = min(max(int(y + sigma), miny + 1), h)
        self.occupancy[f, miny:maxy, minx:maxx] = 1

    def get(self, f, x, y):                                      # occupancy.cpp:32-43
        if f >= self.occupancy.shape[0]:
            return True
        x, y = float(x), float(y)
        if self.reduction != 1.0:
            x /= self.reduction
            y /= self.reduction
        _, h, w = self.occupancy.shape
        xi = min(max(int(x), 0), w - 1)
        yi = min(max(int(y), 0), h - 1)
        return bool(self.occupancy[f, yi, xi])


class TrackAnnotation:
    """The poses one person had in the frames seen so far (reference ``track_annotation.py:4-57``)."""
    track_id_counter = 0

    def __init__(self):
        self.frame_pose = []                       # [(frame number, Annotation)], frame numbers ascending
        TrackAnnotation.track_id_counter += 1
        self.id_ = TrackAnnotation.track_id_counter

    def add(self, frame_number, pose_annotation):
        self.frame_pose.append((frame_number, pose_annotation))
        return self

    def pose(self, frame_number):
        for frame_i, pose in reversed(self.frame_pose):
            if frame_i < frame_number:
                break
            if frame_i == frame_number:
                return pose
        return None

    def pose_score(self, frame_number):
        """Score of the pose in that frame with the tracking weights: nose, eyes and shoulders dominate,
        the last two joints do not count, a pose with fewer than two confident joints scores 0."""
        pose = self.pose(frame_number)
        if pose is None:
            return 0.0
        order = np.argsort(pose.data[:, 2])[::-1]
        if pose.data[order[1], 2] < 0.05:
            return 0.0
        w = pose.score_weights
        w[:] = 1.0
        w[1] = 3.0
        w[2] = 5.0
        w[5:] = 0.1
        w[-2:] = 0.0
        w /= np.sum(w)
        return pose.score

    def score(self, frame_number, current_importance=1.0):
        """Mean pose score over the last 12 frames; ``current_importance`` down-weights the frame that is
        still being processed."""
        weights = [1.0] * 12
        weights[0] = current_importance
        return sum(w * self.pose_score(frame_number - i) for i, w in enumerate(weights)) / sum(weights)

    def __len__(self):
        return len(self.frame_pose)


class TrackBase(Decoder):
    """Track bookkeeping shared by the tracking decoders (reference ``track_base.py:11-168``)."""
    single_pose_threshold = 0.3
    multi_pose_threshold = 0.2
    multi_pose_n = 3
    minimum_threshold = 0.1
    simplify_good_ids = True

    def __init__(self):
        super().__init__()
        self.active: List[TrackAnnotation] = []
        self.frame_number = 0
        self.simplified_track_id_map = {}
        self.simplified_last_track_id = 0

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        group = parser.add_argument_group('Decoder for tracking')
        group.add_argument('--tr-single-pose-threshold', default=cls.single_pose_threshold, type=float,
                           help='Single-pose threshold for tracking.')
        group.add_argument('--tr-multi-pose-threshold', default=cls.multi_pose_threshold, type=float,
                           help='multi-pose threshold for tracking.')
        group.add_argument('--tr-multi-pose-n', default=cls.multi_pose_n, type=float,
                           help='multi-pose n for tracking.')
        group.add_argument('--tr-minimum-threshold', default=cls.minimum_threshold, type=float,
                           help='minimum-pose threshold for tracking.')

    @classmethod
    def configure(cls, args: argparse.Namespace):
        cls.single_pose_threshold = args.tr_single_pose_threshold
        cls.multi_pose_threshold = args.tr_multi_pose_threshold
        cls.multi_pose_n = args.tr_multi_pose_n
        cls.minimum_threshold = args.tr_minimum_threshold

    def reset(self):
        self.active = []
        self.frame_number = 0
        self.simplified_track_id_map = {}
        self.simplified_last_track_id = 0

    def simplify_ids(self, ids):
        out = []
        for id_ in ids:
            if id_ not in self.simplified_track_id_map:
                self.simplified_last_track_id += 1
                self.simplified_track_id_map[id_] = self.simplified_last_track_id
            out.append(self.simplified_track_id_map[id_])
        return out

    def annotations(self, frame_number):
        """Poses of the good tracks seen in this frame, carrying (simplified) track ids (:91-104)."""
        tracks = [t for t in self.active
                  if t.frame_pose[-1][0] == frame_number and self.track_is_good(t, frame_number)]
        if not tracks:
            return []
        ids = [t.id_ for t in tracks]
        if self.simplify_good_ids:
            ids = self.simplify_ids(ids)
        annotations = [t.frame_pose[-1][1] for t in tracks]
        for ann, id_ in zip(annotations, ids):
            ann.id_ = id_
        return annotations

    def track_is_viable(self, track, frame_number):              # :141-148
        if frame_number > track.frame_pose[-1][0] + 33:
            return False
        return any(track.pose_score(frame_number - i) > self.multi_pose_threshold for i in range(33))

    def track_is_good(self, track, frame_number):                # :150-168
        for i in range(4):
            pose = track.pose(frame_number - i)
            if pose is not None and getattr(pose, 'ignore_region', False):
                return False
        if not self.track_is_viable(track, frame_number):
            return False
        recent = [track.pose_score(frame_number - i) for i in range(6)]
        if all(s < self.single_pose_threshold for s in recent) and \
                sum(1 for s in recent if s > self.multi_pose_threshold) < self.multi_pose_n:
            return False
        assert self.minimum_threshold >= 0.0
        return track.pose_score(frame_number) > self.minimum_threshold


class TrackingPose(TrackBase):
    """Reference ``tracking_pose.py:18-300``: heads (single-image CIF, single-image CAF, temporal CAF)."""
    cache_group = [0, -1]
    forward_tracking_pose = True
    track_recovery = False
    single_seed = False

    def __init__(self, cif_meta: headmeta.TSingleImageCif, caf_meta: headmeta.TSingleImageCaf,
                 tcaf_meta: headmeta.Tcaf, *, pose_generator=None):
        super().__init__()
        self.cif_meta, self.caf_meta, self.tcaf_meta = cif_meta, caf_meta, tcaf_meta
        self.priority = 1.0 + (cif_meta.n_fields + caf_meta.n_fields + tcaf_meta.n_fields) / 1000.0
        self.invalid_keypoints = [
            i for i, kp in enumerate(cif_meta.keypoints) if kp in ('left_ear', 'right_ear')
        ] if cif_meta.dataset == 'posetrack2018' else []

        self.n_keypoints = len(cif_meta.keypoints)
        n_frames = len(self.cache_group)
        tracking_keypoints = list(cif_meta.keypoints) * n_frames
        tracking_sigmas = list(cif_meta.sigmas) * n_frames
        tracking_skeleton = list(caf_meta.skeleton) + [              # temporal bones: joint k now -> joint k then
            (k + 1, k + 1 + frame_i * self.n_keypoints)
            for frame_i in range(1, n_frames) for k in range(self.n_keypoints)]
        self.tracking_cif_meta = headmeta.Cif('tracking_cif', cif_meta.dataset, keypoints=tracking_keypoints,
                                              sigmas=tracking_sigmas, pose=None)
        self.tracking_cif_meta.head_index = 0
        self.tracking_cif_meta.base_stride = cif_meta.base_stride
        self.tracking_cif_meta.upsample_stride = cif_meta.upsample_stride
        self.tracking_caf_meta = headmeta.Caf('tracking_caf', caf_meta.dataset, keypoints=tracking_keypoints,
                                              sigmas=tracking_sigmas, skeleton=tracking_skeleton, pose=None)
        self.tracking_caf_meta.head_index = 1
        self.tracking_caf_meta.base_stride = caf_meta.base_stride
        self.tracking_caf_meta.upsample_stride = caf_meta.upsample_stride
        # the decode: n_keypoints = 34 joints, 17 CIF fields -- one association-kernel launch per frame
        self.pose_generator = pose_generator or CifCaf([self.tracking_cif_meta], [self.tracking_caf_meta])
        self.nms_occupancy = Occupancy(2, 4)

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        group = parser.add_argument_group('trackingpose decoder')
        group.add_argument('--trackingpose-track-recovery', default=False, action='store_true')
        group.add_argument('--trackingpose-single-seed', default=False, action='store_true')

    @classmethod
    def configure(cls, args: argparse.Namespace):
        cls.track_recovery = args.trackingpose_track_recovery
        cls.single_seed = args.trackingpose_single_seed

    @classmethod
    def factory(cls, head_metas):
        def matching(cifs, cafs, tcafs):
            return [cls(c, a, t) for c, a, t in zip(cifs, cafs, tcafs)
                    if isinstance(c, headmeta.TSingleImageCif) and isinstance(a, headmeta.TSingleImageCaf)
                    and isinstance(t, headmeta.Tcaf)]
        if len(head_metas) < 3:
            return []
        return matching(head_metas, head_metas[1:], head_metas[2:]) + \
            matching(head_metas, head_metas[1:], head_metas[3:])

    def soft_nms(self, tracks, frame_number):
        """Reference ``:114-160``: suppress joints of weaker tracks that sit on a stronger track's joint."""
        if not tracks:
            return
        kp_th = native.NMSKeypoints.get_keypoint_threshold()
        for t in tracks:
            ann = t.pose(self.frame_number)
            if ann is None:
                continue
            ann.data[ann.data[:, 2] < kp_th] = 0.0
            ann.data[self.invalid_keypoints] = 0.0
        self.nms_occupancy.reset((
            self.n_keypoints,
            int(max(1, max(np.max(t.frame_pose[-1][1].data[:, 1]) for t in tracks) + 1)),
            int(max(1, max(np.max(t.frame_pose[-1][1].data[:, 0]) for t in tracks) + 1)),
        ))
        for track in sorted(tracks, key=lambda tr: -tr.score(frame_number, current_importance=0.01)):
            ann = track.pose(frame_number)
            if ann is None:
                continue
            for joint_i in np.flatnonzero(ann.data[:, 2]):
                xyv = ann.data[joint_i]
                if self.nms_occupancy.get(joint_i, xyv[0], xyv[1]):
                    xyv[2] = 0.0
                else:
                    self.nms_occupancy.set(joint_i, xyv[0], xyv[1], ann.joint_scales[joint_i])
        for t in tracks:
            ann = t.pose(self.frame_number)
            if ann is not None:
                ann.data[ann.data[:, 2] < kp_th] = 0.0

    def _initial_annotations(self):
        """Tracking poses seeded with the active tracks' earlier poses (``:163-202``), tallest first."""
        initial = []
        K = self.n_keypoints
        for track in self.active:
            ann = Annotation(self.tracking_cif_meta.keypoints, self.tracking_caf_meta.skeleton)
            ann.id_ = track.id_
            for position_i, frame_i in enumerate(self.cache_group[1:], start=1):
                prev = track.pose(self.frame_number + frame_i)
                if prev is not None:
                    ann.data[K * position_i:K * (position_i + 1)] = prev.data
                    ann.joint_scales[K * position_i:K * (position_i + 1)] = prev.joint_scales
            if self.single_seed:
                weaker = ann.data[:, 2] < np.amax(ann.data[:, 2])
                ann.data[weaker] = 0.0
                ann.joint_scales[weaker] = 0.0
            ann.data[ann.data[:, 2] < 0.05] = 0.0
            if np.any(ann.data[:, 2] > 0.0):
                initial.append(ann)
        return sorted(initial, key=lambda a: a.bbox()[3], reverse=True)

    def __call__(self, fields, *, initial_annotations=None):
        self.frame_number += 1
        start = time.perf_counter()
        initial = self._initial_annotations()
        tracking_fields = [
            fields[self.cif_meta.head_index],
            torch.cat([fields[self.caf_meta.head_index], fields[self.tcaf_meta.head_index]], dim=0),
        ]
        tracking_annotations = self.pose_generator(tracking_fields, initial_annotations=initial)

        # the first K joints of every tracking pose are this frame's pose of that track (:220-243)
        active_by_id = {t.id_: t for t in self.active}
        lost = {t.id_: t.frame_pose[-1][0] for t in self.active if t.frame_pose[-1][0] < self.frame_number - 1}
        K = self.n_keypoints
        for tracking_ann in tracking_annotations:
            ann = Annotation(self.cif_meta.keypoints, self.caf_meta.skeleton)
            ann.data[:] = tracking_ann.data[:K]
            ann.joint_scales = tracking_ann.joint_scales[:K]
            track_id = getattr(tracking_ann, 'id_', -1)
            if track_id == -1:
                new_track = TrackAnnotation().add(self.frame_number, ann)
                self.active.append(new_track)
                tracking_ann.id_ = new_track.id_
            else:
                active_by_id[track_id].add(self.frame_number, ann)

        self.soft_nms(self.active, self.frame_number)

        if self.track_recovery:                                  # :248-266
            removed = set()
            for track in self.active:
                if not lost:
                    break
                if len(track) > 1 or track.pose(self.frame_number) is None:
                    continue
                track_id = max(lost.items(), key=lambda d: d[1])[0]
                del lost[track_id]
                active_by_id[track_id].add(self.frame_number, track.pose(self.frame_number))
                removed.add(track)
                LOG.info('recovered track %d', track_id)
            self.active = [t for t in self.active if t not in removed]

        self.active = [t for t in self.active if self.track_is_viable(t, self.frame_number)]
        LOG.debug('track time: %.3fs, active tracks = %d', time.perf_counter() - start, len(self.active))
        return self.annotations(self.frame_number)


# ----------------------------------------------------------------------------- PoseSimilarity
def _track_pose_for(track, frame_number, track_frame):
    """Common preamble of the reference's distance functions (``pose_distance/euclidean.py:24-37``): which
    stored pose of the track to compare with, or None when the track was out of sight for too long."""
    skipped_frames = frame_number - track.frame_pose[-1][0] - 1
    assert skipped_frames >= 0
    if skipped_frames > 12:
        return None, skipped_frames
    track_frame += skipped_frames
    if track_frame > -1 or len(track.frame_pose) < -track_frame:
        return None, skipped_frames
    return track.frame_pose[track_frame][1], track_frame


class Euclidean:
    """Mean keypoint distance between a new pose and a track's pose ``track_frames`` back
    (reference ``pose_distance/euclidean.py:4-46``)."""
    invisible_penalty = 110.0

    def __init__(self, *, track_frames=None):
        self.track_frames = [-1] if track_frames is None else track_frames
        assert all(t < 0 for t in self.track_frames)
        self.valid_keypoints = None

    def __call__(self, frame_number, pose, track, track_is_good):
        return min(self.distance(frame_number, pose, track, track_is_good, t) for t in self.track_frames)

    def distance(self, frame_number, pose, track, track_is_good, track_frame=-1):
        other, _ = _track_pose_for(track, frame_number, track_frame)
        if other is None:
            return 1000.0
        pose1, pose2 = pose.data[self.valid_keypoints], other.data[self.valid_keypoints]
        d = np.clip(np.linalg.norm(pose2[:, :2] - pose1[:, :2], axis=1), 0.0, self.invisible_penalty)
        d[pose1[:, 2] < 0.05] = self.invisible_penalty
        d[pose2[:, 2] < 0.05] = self.invisible_penalty
        return np.mean(d)


class Oks:
    """110 * (1 - object keypoint similarity) (reference ``pose_distance/oks.py:4-68``)."""
    inflate = 1.0

    def __init__(self, *, track_frames=None):
        self.track_frames = [-1] if track_frames is None else track_frames
        assert all(t < 0 for t in self.track_frames)
        self.valid_keypoints = None
        self.sigmas = None

    def __call__(self, frame_number, pose, track, track_is_good):
        return min(self.distance(frame_number, pose, track, track_is_good, t) for t in self.track_frames)

    @staticmethod
    def scale(pose):
        pose = pose[pose[:, 2] > 0.0]
        return np.sqrt((pose[:, 0].max() - pose[:, 0].min()) * (pose[:, 1].max() - pose[:, 1].min()))

    def distance(self, frame_number, pose, track, track_is_good, track_frame=-1):
        other, _ = _track_pose_for(track, frame_number, track_frame)
        if other is None:
            return 1000.0
        pose1, pose2 = pose.data[self.valid_keypoints], other.data[self.valid_keypoints]
        visible = np.logical_and(pose1[:, 2] > 0.0, pose2[:, 2] > 0.0)
        if not np.any(visible):
            return 1000.0
        scale = max(1.0, 0.5 * (self.scale(pose1) + self.scale(pose2)))
        d = np.linalg.norm(pose2[:, :2] - pose1[:, :2], axis=1)
        k = 2.0 * self.sigmas[self.valid_keypoints] * self.inflate
        g = np.exp(-0.5 * d ** 2 / (scale ** 2 * k ** 2))
        return 110.0 * (1.0 - np.mean(g[visible]))


class Crafted:
    """Hand-crafted distance (reference ``pose_distance/crafted.py:7-90``): centred keypoint distance plus
    penalties for young / bad tracks, weak poses and skipped frames.  (The reference calls ``pose.score()``
    although ``score`` is a property, ``crafted.py:74-77``, so its version raises as soon as it gets
    that far; this one reads the property.)"""
    invisible_penalty = 110.0

    def __init__(self):
        self.valid_keypoints = None

    def __call__(self, frame_number, pose, track, track_is_good):
        return min(self.distance(frame_number, pose, track, track_is_good, t) for t in (-1, -4, -8, -12))

    def distance(self, frame_number, pose, track, track_is_good, track_frame=-1):
        other, track_frame = _track_pose_for(track, frame_number, track_frame)
        if other is None:
            return 1000.0
        pose1, pose2 = pose.data[self.valid_keypoints], other.data[self.valid_keypoints]
        best = np.argsort(pose1[:, 2] * pose2[:, 2])[::-1]
        if pose1[best[2], 2] < 0.05 or pose2[best[2], 2] < 0.05:
            return 1000.0
        c1, c2 = np.mean(pose1[best[:3], :2], axis=0), np.mean(pose2[best[:3], :2], axis=0)
        d = np.linalg.norm((pose2[:, :2] - c2) - (pose1[:, :2] - c1), axis=1)
        d = np.clip(d, 0.0, self.invisible_penalty)
        d[pose1[:, 2] < 0.05] = self.invisible_penalty
        d[pose2[:, 2] < 0.05] = self.invisible_penalty
        track_penalty = 40.0 if len(track.frame_pose) < 4 else (8.0 if len(track.frame_pose) < 8 else 0.0)
        if not track_is_good:
            track_penalty = max(track_penalty, 8.0)
        pose_penalty = 40.0 if pose.score < 0.2 else (8.0 if pose.score < 0.5 else 0.0)
        skipped_frame_cost = 40.0 if track_frame < -1 else 0.0
        return np.linalg.norm(c2 - c1) / 10.0 + np.mean(d) + track_penalty + pose_penalty + skipped_frame_cost


class PoseSimilarity(TrackBase):
    """Tracking by matching single-frame poses (one HIP decode per frame) to the active tracks with the
    Hungarian algorithm over a pose distance (reference ``decoder/pose_similarity.py:20-141``)."""
    distance_type = Euclidean

    def __init__(self, cif_meta: headmeta.Cif, caf_meta: headmeta.Caf, *, pose_generator=None):
        super().__init__()
        self.cif_meta, self.caf_meta = cif_meta, caf_meta
        self.priority = -10.0 + (cif_meta.n_fields + caf_meta.n_fields) / 1000.0
        self.distance_function = self.distance_type()
        skip = ('left_ear', 'right_ear') if cif_meta.dataset == 'posetrack2018' else ()
        self.distance_function.valid_keypoints = [i for i, kp in enumerate(cif_meta.keypoints) if kp not in skip]
        self.distance_function.sigmas = np.asarray(cif_meta.sigmas)
        self.pose_generator = pose_generator or CifCaf([cif_meta], [caf_meta])

    @classmethod
    def cli(cls, parser: argparse.ArgumentParser):
        group = parser.add_argument_group('PoseSimilarity')
        group.add_argument('--posesimilarity-distance', default='euclidean',
                           choices=('crafted', 'euclidean', 'euclidean4', 'oks'))
        group.add_argument('--posesimilarity-oks-inflate', default=Oks.inflate, type=float)

    @classmethod
    def configure(cls, args: argparse.Namespace):
        choice = args.posesimilarity_distance
        if choice == 'euclidean':
            cls.distance_type = Euclidean
        elif choice == 'euclidean4':
            cls.distance_type = lambda _=None: Euclidean(track_frames=[-1, -4, -8, -12])
        elif choice == 'oks':
            cls.distance_type = Oks
        elif choice == 'crafted':
            cls.distance_type = Crafted
        else:
            raise RuntimeError('distance function type not known')
        Oks.inflate = args.posesimilarity_oks_inflate

    @classmethod
    def factory(cls, head_metas):
        if len(head_metas) < 2:
            return []
        return [cls(cif_meta, caf_meta) for cif_meta, caf_meta in zip(head_metas, head_metas[1:])
                if isinstance(cif_meta, headmeta.Cif) and isinstance(caf_meta, headmeta.Caf)]

    def prune_active(self, frame_number):                        # track_base.py:85-89
        self.active = [t for t in self.active if frame_number - t.frame_pose[-1][0] <= 33]
        self.active = [t for t in self.active
                       if frame_number - t.frame_pose[-1][0] == 1 or len(t.frame_pose) > 2]

    def __call__(self, fields, *, initial_annotations=None):
        import scipy.optimize
        self.frame_number += 1
        self.prune_active(self.frame_number)
        poses = self.pose_generator(fields)

        # rows: every active track, then one "track is lost this frame" row per track at a flat cost
        n_tracks = len(self.active)
        cost = np.full((n_tracks * 2, len(poses)), 1000.0)
        for track_i, track in enumerate(self.active):
            good = self.track_is_good(track, self.frame_number)
            for pose_i, pose in enumerate(poses):
                cost[track_i, pose_i] = self.distance_function(self.frame_number, pose, track, good)
                cost[track_i + n_tracks, pose_i] = 100.0
        track_indices, pose_indices = scipy.optimize.linear_sum_assignment(cost)
        matched = set()
        for track_i, pose_i in zip(track_indices, pose_indices):
            if track_i >= n_tracks:
                continue
            self.active[track_i].add(self.frame_number, poses[pose_i])
            matched.add(pose_i)
        for pose_i, pose in enumerate(poses):
            if pose_i not in matched:
                self.active.append(TrackAnnotation().add(self.frame_number, pose))

        self.active = [t for t in self.active if self.track_is_viable(t, self.frame_number)]
        return self.annotations(self.frame_number)


_decoder.DECODERS.add(TrackingPose)
_decoder.DECODERS.add(PoseSimilarity)


# ---------------------------------------------------------------- a host that has the reference package itself
_HOST_CLASSES = {}


def host_classes(openpifpaf):
    """-> ``(TrackingPose, PoseSimilarity)`` for a host process that HAS the reference package (``openpifpaf_amd.register()``
    running as its plugin): the reference's OWN classes -- track bookkeeping, soft NMS between tracks, pruning rules,
    the pose-similarity matching are its code, not this module's restatement of it -- subclassed for one thing only: the
    pose generator they decode with is the HIP ``CifCaf`` (reference ``decoder/tracking_pose.py:26-83``,
    ``decoder/pose_similarity.py:23-42`` take it as ``pose_generator=``).  The classes above stay what the package uses
    where the reference is not importable (the GPU box, a stand-alone installation); both are checked against the same
    golden videos (``tests/test_tracking_pose.py``)."""
    key = id(openpifpaf)
    if key in _HOST_CLASSES:
        return _HOST_CLASSES[key]
    from openpifpaf.decoder import pose_similarity as ref_ps, tracking_pose as ref_tp

    class _NoPoseGenerator:                       # (keeps the reference constructor from building ITS CifCaf, which needs its extension)
        occupancy_visualizer = None

    class HostTrackingPose(ref_tp.TrackingPose):
        """The reference's ``TrackingPose`` decoding through ``openpifpaf_amd.decoder.CifCaf``."""

        def __init__(self, cif_meta, caf_meta, tcaf_meta, *, pose_generator=None):
            super().__init__(cif_meta, caf_meta, tcaf_meta, pose_generator=pose_generator or _NoPoseGenerator())
            if pose_generator is None:
                self.pose_generator = CifCaf([self.tracking_cif_meta], [self.tracking_caf_meta])

        def __call__(self, fields, *, initial_annotations=None):
            CifCaf.nms = None                     # (what the reference does to its own class: tracking_pose.py:203-205)
            return super().__call__(fields, initial_annotations=initial_annotations)

    class HostPoseSimilarity(ref_ps.PoseSimilarity):
        """The reference's ``PoseSimilarity`` decoding through ``openpifpaf_amd.decoder.CifCaf``."""

        def __init__(self, cif_meta, caf_meta, *, pose_generator=None):
            super().__init__(cif_meta, caf_meta, pose_generator=pose_generator or _NoPoseGenerator())
            if pose_generator is None:
                self.pose_generator = CifCaf([cif_meta], [caf_meta])

    HostTrackingPose.__name__ = HostTrackingPose.__qualname__ = 'TrackingPose'
    HostPoseSimilarity.__name__ = HostPoseSimilarity.__qualname__ = 'PoseSimilarity'
    _HOST_CLASSES[key] = (HostTrackingPose, HostPoseSimilarity)
    return _HOST_CLASSES[key]

