"""openpifpaf_amd.tracking.TrackingPose against the reference's own TrackingPose (golden: its Python + its
C++ decoder run over seeded synthetic videos, tests/golden/make_golden_tracking_pose.py)."""
import os

import numpy as np
import pytest

from common import TRACKING_VIDEOS

torch = pytest.importorskip('torch')
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tracking_pose_golden.npz')


def metas():
    from openpifpaf_amd import constants, headmeta
    cif = headmeta.TSingleImageCif('cif', 'synthetic', keypoints=constants.COCO_KEYPOINTS,
                                   sigmas=constants.COCO_PERSON_SIGMAS, pose=constants.COCO_UPRIGHT_POSE,
                                   draw_skeleton=constants.COCO_PERSON_SKELETON)
    caf = headmeta.TSingleImageCaf('caf', 'synthetic', keypoints=constants.COCO_KEYPOINTS,
                                   sigmas=constants.COCO_PERSON_SIGMAS, pose=constants.COCO_UPRIGHT_POSE,
                                   skeleton=constants.COCO_PERSON_SKELETON)
    tcaf = headmeta.Tcaf('tcaf', 'synthetic', keypoints_single_frame=constants.COCO_KEYPOINTS,
                         sigmas_single_frame=constants.COCO_PERSON_SIGMAS, pose_single_frame=constants.COCO_UPRIGHT_POSE)
    for i, m in enumerate((cif, caf, tcaf)):
        m.head_index = i
        m.base_stride = 16
        m.upsample_stride = 2
    return cif, caf, tcaf


class OraclePoseGenerator:
    """Stands in for decoder.CifCaf in the CPU test: same call signature, the decode done by the oracle."""

    def __init__(self, keypoints, skeleton):
        self.keypoints, self.skeleton = keypoints, skeleton
        self.skeleton0 = np.asarray(skeleton, dtype=np.int64) - 1

    def __call__(self, fields, initial_annotations=None):
        from openpifpaf_amd.annotation import Annotation
        from oracle import port
        init = ids = None
        if initial_annotations:
            init = np.zeros((len(initial_annotations), len(self.keypoints), 4), dtype=np.float32)
            ids = np.zeros((len(initial_annotations),), dtype=np.int64)
            for i, a in enumerate(initial_annotations):
                init[i, :, 0], init[i, :, 1:3], init[i, :, 3] = a.data[:, 2], a.data[:, :2], a.joint_scales
                ids[i] = getattr(a, 'id_', -1)
        out, out_ids = port.decode(fields[0].numpy(), 8, fields[1].numpy(), 8, self.skeleton0,
                                   n_keypoints=len(self.keypoints), initial_annotations=init, initial_ids=ids)
        anns = []
        for data, id_ in zip(out, out_ids):
            ann = Annotation(self.keypoints, self.skeleton)
            ann.data[:, :2], ann.data[:, 2], ann.joint_scales[:] = data[:, 1:3], data[:, 0], data[:, 3]
            if id_ != -1:
                ann.id_ = int(id_)
            anns.append(ann)
        return anns


def run_video(make_tracker, video, golden, v, to_device):
    from openpifpaf_amd import synth, tracking
    seed, people, n_frames, appear = video
    tracking.TrackAnnotation.track_id_counter = 0
    tracker = make_tracker()
    for t, fields in enumerate(synth.synth_tracking_sequence(seed, people, n_frames, appear=appear)):
        anns = tracker([to_device(torch.from_numpy(f)) for f in fields])
        want_ids = golden['video%d_frame%d_ids' % (v, t)]
        want = golden['video%d_frame%d_data' % (v, t)]
        assert [a.id_ for a in anns] == want_ids.tolist(), 'video %d frame %d' % (v, t)
        got = np.asarray([a.data for a in anns], dtype=np.float32).reshape(-1, 17, 3)
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4, 'video %d frame %d' % (v, t)
        scales = np.asarray([a.joint_scales for a in anns], dtype=np.float32).reshape(-1, 17)
        assert np.abs(scales - golden['video%d_frame%d_scales' % (v, t)]).max() <= 1e-4
    return tracker


@pytest.mark.parametrize('v', range(len(TRACKING_VIDEOS)))
def test_host_logic_with_oracle_decode_matches_reference_tracker(v):
    from openpifpaf_amd import tracking
    golden = np.load(GOLDEN)

    def make():
        cif, caf, tcaf = metas()
        keypoints = list(cif.keypoints) * 2
        skeleton = list(caf.skeleton) + [(k + 1, k + 18) for k in range(17)]
        return tracking.TrackingPose(cif, caf, tcaf, pose_generator=OraclePoseGenerator(keypoints, skeleton))

    tracker = run_video(make, TRACKING_VIDEOS[v], golden, v, lambda t: t)
    assert tracker.frame_number == TRACKING_VIDEOS[v][2] and len(tracker.active) >= 1


@pytest.mark.parametrize('name', ['euclidean', 'oks'])
@pytest.mark.parametrize('v', range(len(TRACKING_VIDEOS)))
def test_pose_similarity_with_oracle_decode_matches_reference(v, name):
    from openpifpaf_amd import synth, tracking
    golden = np.load(GOLDEN)
    cif, caf, _ = metas()
    tracking.TrackAnnotation.track_id_counter = 0
    tracking.PoseSimilarity.distance_type = {'euclidean': tracking.Euclidean, 'oks': tracking.Oks}[name]
    try:
        tracker = tracking.PoseSimilarity(cif, caf, pose_generator=OraclePoseGenerator(cif.keypoints, caf.skeleton))
        seed, people, n_frames, appear = TRACKING_VIDEOS[v]
        for t, fields in enumerate(synth.synth_tracking_sequence(seed, people, n_frames, appear=appear)):
            anns = tracker([torch.from_numpy(f) for f in fields[:2]])
            assert [a.id_ for a in anns] == golden['sim_%s_video%d_frame%d_ids' % (name, v, t)].tolist()
            got = np.asarray([a.data for a in anns], dtype=np.float32).reshape(-1, 17, 3)
            # the reference tracker keeps ONE native decoder for the whole video, whose results drift with
            # the call count (float32 revision offset): tolerance, not equality
            want = golden['sim_%s_video%d_frame%d_data' % (name, v, t)]
            assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4
    finally:
        tracking.PoseSimilarity.distance_type = tracking.Euclidean


def test_crafted_distance_prefers_the_same_person():
    from openpifpaf_amd import tracking
    from openpifpaf_amd.annotation import Annotation
    cif, caf, _ = metas()

    def pose(dx):
        a = Annotation(cif.keypoints, caf.skeleton)
        a.data[:, 0] = np.arange(17) * 3.0 + dx
        a.data[:, 1] = np.arange(17) * 2.0
        a.data[:, 2] = 0.8
        return a
    d = tracking.Crafted()
    d.valid_keypoints = list(range(17))
    track = tracking.TrackAnnotation().add(1, pose(0.0))
    near, far = d(2, pose(1.0), track, True), d(2, pose(60.0), track, True)
    assert near < far < 1000.0 and d(20, pose(0.0), track, True) == 1000.0      # out of sight for > 12 frames


def test_factory_and_registry():
    from openpifpaf_amd import decoder, tracking
    assert tracking.TrackingPose in decoder.DECODERS and tracking.PoseSimilarity in decoder.DECODERS
    assert tracking.TrackingPose.factory(list(metas())[:2]) == []          # needs the three tracking heads
    import argparse
    parser = argparse.ArgumentParser()
    decoder.cli(parser)
    args = parser.parse_args(['--trackingpose-single-seed', '--tr-minimum-threshold', '0.2'])
    try:
        decoder.configure(args)
        assert tracking.TrackingPose.single_seed is True and tracking.TrackBase.minimum_threshold == 0.2
    finally:
        from openpifpaf_amd import _lib
        _lib.set_params(_lib.default_params())
        decoder.Factory.decoder_request = None
        tracking.TrackingPose.single_seed = False
        tracking.TrackBase.minimum_threshold = 0.1


@pytest.mark.gpu
@pytest.mark.parametrize('v', range(len(TRACKING_VIDEOS)))
def test_tracking_pose_on_the_hip_path_matches_reference_tracker(v):
    from openpifpaf_amd import tracking
    golden = np.load(GOLDEN)

    def make():
        decs = tracking.TrackingPose.factory(list(metas()))
        assert len(decs) == 1 and decs[0].pose_generator.cpp_decoder.n_keypoints == 34
        return decs[0]

    run_video(make, TRACKING_VIDEOS[v], golden, v, lambda t: t.cuda())
