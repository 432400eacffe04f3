"""The restatement against the REAL reference decoder (oracle/_ref, built from the
reference's own sources).  Skipped where _ref is unavailable (it needs /root/reference
to build); the committed golden vectors pin the oracle everywhere else."""
import numpy as np
import pytest

from oracle import reference

pytestmark = pytest.mark.skipif(not reference.available(), reason='oracle/_ref/openpifpaf_ref.so not built')


@pytest.fixture(scope='module')
def ref():
    torch = reference.load()
    torch.set_num_threads(1)
    reference.reset_statics()
    yield reference
    reference.reset_statics()


@pytest.mark.parametrize('seed,people', [(100, 1), (101, 4), (102, 9), (103, 16)])
def test_decode_bit_equal(ref, coco_skeleton0, seed, people):
    from openpifpaf_amd import synth
    from oracle import port
    cif, caf = synth.synth_fields(seed, people)
    r_out, r_ids, r_hr = ref.decode(cif, 8, caf, 8, coco_skeleton0)
    o_out, o_ids, o_hr = port.decode(cif, 8, caf, 8, coco_skeleton0, return_cifhr=True)
    assert np.array_equal(r_hr, o_hr)
    assert r_out.shape == o_out.shape and np.array_equal(r_out, o_out)
    assert np.array_equal(r_ids, o_ids)


@pytest.mark.parametrize('kw', [
    dict(greedy=1), dict(reverse_match=0), dict(keypoint_threshold=0.3, keypoint_threshold_rel=0.7),
    dict(cif_threshold=0.2, seed_threshold=0.3, caf_threshold=0.25),
    dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
         nms_instance_threshold=0.0, nms_keypoint_threshold=0.0),
    dict(ablation_cifseeds_nms=1), dict(ablation_cifseeds_no_rescore=1), dict(ablation_caf_no_rescore=1),
])
def test_decode_with_options(ref, coco_skeleton0, kw):
    from openpifpaf_amd import synth
    from oracle import port
    cif, caf = synth.synth_fields(200, 6)
    p = port.default_params(**kw)
    ref.apply_params(p)
    try:
        r_out, _, _ = ref.decode(cif, 8, caf, 8, coco_skeleton0)
    finally:
        ref.reset_statics()
    o_out, _ = port.decode(cif, 8, caf, 8, coco_skeleton0, params=p)
    assert r_out.shape == o_out.shape and np.array_equal(r_out, o_out), kw


@pytest.mark.parametrize('seed,people,size', [(0, 2, 41), (1, 4, 81), (2, 7, 97)])
def test_wholebody_133_keypoints_bit_equal(ref, seed, people, size):
    """BASELINE config 4 shapes (133 keypoints, 160 bones, joint degree up to 6): the restatement against the
    real decoder, normal and force-complete."""
    from openpifpaf_amd import constants, synth
    from oracle import port
    wb = constants.wholebody()
    skel0 = np.asarray(wb['skeleton'], dtype=np.int64) - 1
    cif, caf = synth.synth_fields(seed, people, height=size, width=size, pose=wb['standing_pose'],
                                  skeleton=wb['skeleton'], size_range=(0.6, 0.95))
    r_out, r_ids, r_hr = ref.decode(cif, 8, caf, 8, skel0)
    o_out, o_ids, o_hr = port.decode(cif, 8, caf, 8, skel0, return_cifhr=True)
    assert len(r_out) >= 1 and np.array_equal(r_hr, o_hr)
    assert r_out.shape == o_out.shape and np.array_equal(r_out, o_out) and np.array_equal(r_ids, o_ids)
    p = port.default_params(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                            nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
    ref.apply_params(p)
    try:
        r_fc, _, _ = ref.decode(cif, 8, caf, 8, skel0)
    finally:
        ref.reset_statics()
    o_fc, _ = port.decode(cif, 8, caf, 8, skel0, params=p)
    assert r_fc.shape == o_fc.shape and np.array_equal(r_fc, o_fc)


@pytest.mark.parametrize('seed,people', [(81, 3), (82, 7), (83, 12)])
def test_dense_connections_44_bones_bit_equal(ref, seed, people):
    """CifCafDense's decoder input (reference decoder/cifcaf.py:17-78): 19 sparse + 25 dense bones as one CAF
    head, so a joint has many competing incoming connections."""
    from openpifpaf_amd import constants, synth
    from oracle import port
    skeleton = list(constants.COCO_PERSON_SKELETON) + list(constants.DENSER_COCO_PERSON_CONNECTIONS)
    skel0 = np.asarray(skeleton, dtype=np.int64) - 1
    cif, caf = synth.synth_fields(seed, people, height=65, width=65, skeleton=skeleton)
    assert caf.shape[0] == 44
    r_out, r_ids, _ = ref.decode(cif, 8, caf, 8, skel0)
    o_out, o_ids = port.decode(cif, 8, caf, 8, skel0)
    assert len(r_out) >= 1 and r_out.shape == o_out.shape
    assert np.array_equal(r_out, o_out) and np.array_equal(r_ids, o_ids)


def test_cifdetseeds_and_occupancy_classes(ref):
    """The two remaining openpifpaf_decoder_utils classes (module.cpp:67-73,96-102): the oracle's CifDetSeeds
    restatement and the binding's host-side Occupancy against the real ones."""
    from openpifpaf_amd import synth, torchscript
    from oracle import port
    torch = ref.load()
    RU = torch.classes.openpifpaf_decoder_utils
    for seed, n_obj, size in ((3, 5, 33), (4, 9, 49)):
        field = synth.synth_det_field(seed, n_obj, height=size, width=size + 8)
        _, _, _, hr = port.cifdet_decode(field, 8, return_cifhr=True)
        hr_t = torch.from_numpy(hr)               # the real class keeps only an accessor: the tensor must outlive it
        seeds = RU.CifDetSeeds(hr_t, 1.0)
        seeds.fill(torch.from_numpy(field), 8)
        r_f, r_v = seeds.get()
        o_f, o_v = port.cifdetseeds(field, 8, hr)
        assert len(o_f) > 0 and np.array_equal(r_f.numpy(), o_f) and np.array_equal(r_v.numpy(), o_v)
    torchscript.load()
    rng = np.random.default_rng(11)
    for reduction, min_scale in ((2.0, 4.0), (1.0, 0.1)):
        mine = torch.classes.openpifpaf_amd_decoder_utils.Occupancy(reduction, min_scale)
        real = RU.Occupancy(reduction, min_scale)
        for shape in ((3, 41, 57), (4, 18, 18), (3, 41, 57)):
            mine.reset(list(shape)); real.reset(list(shape))
            for _ in range(2):
                for _ in range(20):
                    f = int(rng.integers(shape[0]))
                    x, y, sigma = rng.uniform(-8, shape[2] + 8), rng.uniform(-8, shape[1] + 8), rng.uniform(0.0, 9.0)
                    mine.set(f, x, y, sigma); real.set(f, x, y, sigma)
                for _ in range(200):
                    f = int(rng.integers(shape[0] + 2))
                    x, y = rng.uniform(-8, shape[2] + 8), rng.uniform(-8, shape[1] + 8)
                    assert mine.get(f, x, y) == real.get(f, x, y), (reduction, shape, f, x, y)
                mine.clear(); real.clear()


def test_initial_annotations(ref, coco_skeleton0):
    from openpifpaf_amd import synth
    from oracle import port
    cif, caf = synth.synth_fields(300, 4)
    first, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
    init = first[:2].copy()
    init[:, 5:] = 0.0
    ids = np.array([3, 11], dtype=np.int64)
    r_out, r_ids, _ = ref.decode(cif, 8, caf, 8, coco_skeleton0, initial_annotations=init, initial_ids=ids)
    o_out, o_ids = port.decode(cif, 8, caf, 8, coco_skeleton0, initial_annotations=init, initial_ids=ids)
    assert np.array_equal(r_out, o_out) and np.array_equal(r_ids, o_ids)


def test_keypoint_nms_in_isolation(ref, coco_skeleton0):
    """NMSKeypoints::call alone (empty fields, crafted initial annotations; tests/common.py::nms_cases): the restatement
    against the real reference, bit for bit -- the same cases the HIP path is checked on against the restatement
    (tests/test_gpu_parity_r2.py::test_keypoint_nms_in_isolation)."""
    from common import nms_cases
    from oracle import port
    cif = np.zeros((17, 5, 41, 41), dtype=np.float32)
    caf = np.zeros((19, 8, 41, 41), dtype=np.float32)
    cases, settings = nms_cases()
    removed = 0
    for init in cases:
        ids = np.arange(100, 100 + len(init), dtype=np.int64)
        for kw in settings:
            p = port.default_params(**kw)
            ref.apply_params(p)
            try:
                r_out, r_ids, _ = ref.decode(cif, 8, caf, 8, coco_skeleton0, initial_annotations=init, initial_ids=ids)
            finally:
                ref.reset_statics()
            o_out, o_ids = port.decode(cif, 8, caf, 8, coco_skeleton0, params=p, initial_annotations=init, initial_ids=ids)
            assert r_out.shape == o_out.shape and np.array_equal(r_out, o_out), kw
            assert np.array_equal(r_ids, o_ids), kw
            removed += len(init) - len(o_out)
    assert removed > 10          # the cases do exercise suppression and removal


def test_all_active_adversarial(ref, coco_skeleton0):
    from openpifpaf_amd import synth
    from oracle import port
    cif, caf = synth.adversarial_fields(1, height=21, width=21)
    r_out, _, r_hr = ref.decode(cif, 8, caf, 8, coco_skeleton0)
    o_out, _, o_hr = port.decode(cif, 8, caf, 8, coco_skeleton0, return_cifhr=True)
    assert np.array_equal(r_hr, o_hr) and np.array_equal(r_out, o_out)


def test_register_swaps_decoders_inside_the_real_reference_package():
    """The reference's own Python package (imported with oracle/_ref as its extension): after
    ``openpifpaf_amd.register()`` the set its decoder factory iterates holds the HIP-backed classes."""
    from oracle import reference_python
    import os
    if not os.path.isdir(reference_python.REF_SRC):
        pytest.skip('reference sources not present')
    opp = reference_python.load()
    import openpifpaf_amd
    from openpifpaf_amd import decoder, tracking
    import sys
    ref_factory = sys.modules['openpifpaf.decoder.factory']
    before = set(opp.DECODERS)
    try:
        openpifpaf_amd.register()
        assert ref_factory.DECODERS is opp.DECODERS
        assert {decoder.CifCaf, decoder.CifCafDense, decoder.CifDet} <= ref_factory.DECODERS
        # the tracking decoders of a host that has the reference package: ITS classes (bookkeeping, soft NMS, pruning are the
        # reference's own code), subclassed so that they decode through the HIP CifCaf
        host_tp, host_ps = tracking.host_classes(opp)
        assert {host_tp, host_ps} <= ref_factory.DECODERS
        assert issubclass(host_tp, sys.modules['openpifpaf.decoder.tracking_pose'].TrackingPose)
        assert issubclass(host_ps, sys.modules['openpifpaf.decoder.pose_similarity'].PoseSimilarity)
        assert not [d for d in ref_factory.DECODERS if d.__module__.startswith('openpifpaf.')]
    finally:
        opp.DECODERS.clear()
        opp.DECODERS.update(before)


def test_annotation_objects_equal_the_reference_package():
    """openpifpaf_amd.annotation.Annotation against the reference's own class: score, bbox, json_data and
    inverse_transform on random poses."""
    from oracle import reference_python
    import os
    if not os.path.isdir(reference_python.REF_SRC):
        pytest.skip('reference sources not present')
    opp = reference_python.load()
    from openpifpaf_amd import constants
    from openpifpaf_amd.annotation import Annotation
    rng = np.random.default_rng(11)
    meta = {'offset': np.array((-7.0, 12.0)), 'scale': np.array((0.8, 0.75)), 'hflip': True,
            'rotation': {'angle': 0.0, 'width': None, 'height': None}, 'width_height': np.array((500, 375)),
            'valid_area': np.array((0.0, 0.0, 499.0, 374.0))}
    for _ in range(5):
        data = rng.uniform(0, 200, (17, 3)).astype(np.float32)
        data[:, 2] = rng.uniform(0, 1, 17) * (rng.random(17) > 0.3)
        scales = rng.uniform(1, 9, 17).astype(np.float32)
        pair = []
        for cls in (Annotation, opp.Annotation):
            a = cls(constants.COCO_KEYPOINTS, constants.COCO_PERSON_SKELETON,
                    score_weights=constants.COCO_PERSON_SCORE_WEIGHTS)
            a.data[:] = data
            a.joint_scales[:] = scales
            pair.append(a)
        mine, ref = pair
        assert np.isclose(mine.score, ref.score) and np.allclose(mine.bbox(), ref.bbox())
        assert mine.json_data() == ref.json_data()
        m2, r2 = mine.inverse_transform(meta), ref.inverse_transform(meta)
        assert np.allclose(m2.data, r2.data, atol=1e-5) and np.allclose(m2.joint_scales, r2.joint_scales)


def test_composite_field_head_equals_the_reference_package():
    """network.CompositeField4 (conv 1x1 -> PixelShuffle -> crop -> sigmoid / cell offsets / softplus) against
    the reference's ``network/heads.py:272-378`` with identical weights: the field layout the decoder reads."""
    from oracle import reference_python
    import os
    if not os.path.isdir(reference_python.REF_SRC):
        pytest.skip('reference sources not present')
    opp = reference_python.load()
    import torch
    from openpifpaf_amd import headmeta, network
    torch.manual_seed(0)
    mine_metas = headmeta.cocokp_metas()
    ref_cif = opp.headmeta.Cif('cif', 'cocokp', keypoints=mine_metas[0].keypoints, sigmas=mine_metas[0].sigmas,
                               pose=np.asarray(mine_metas[0].pose), draw_skeleton=mine_metas[1].skeleton)
    ref_caf = opp.headmeta.Caf('caf', 'cocokp', keypoints=mine_metas[0].keypoints, sigmas=mine_metas[0].sigmas,
                               pose=np.asarray(mine_metas[0].pose), skeleton=mine_metas[1].skeleton)
    for m in (ref_cif, ref_caf):
        m.base_stride, m.upsample_stride = 16, 2
    x = torch.randn(2, 64, 7, 9)
    for mine_meta, ref_meta in zip(mine_metas, (ref_cif, ref_caf)):
        mine = network.CompositeField4(mine_meta, 64).eval()
        ref = opp.network.heads.CompositeField4(ref_meta, 64).eval()
        ref.conv.load_state_dict(mine.conv.state_dict())
        with torch.no_grad():
            a, b = mine(x), ref(x)
        assert a.shape == b.shape == (2, mine_meta.n_fields, 5 if mine_meta is mine_metas[0] else 8, 13, 17)
        assert torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())


@pytest.mark.parametrize('case', [(9001, 40, 161, 161, (0.2, 0.6)), (9002, 25, 121, 161, (0.3, 0.8)),
                                  (9005, 90, 161, 161, (0.2, 0.5))])
def test_large_fields_bit_equal(ref, coco_skeleton0, case):
    """The 1281-px cases of tests/test_gpu_large_fields.py (161 x 161 and 121 x 161 fields, up to 90 people, more than
    8192 seeds): the restatement the GPU tests compare with is bit-identical to the real reference there too -- default
    flags, the reference benchmark's force-complete setting, and bf16-rounded fields (tied scores: std::sort's order)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import to_bf16
    from openpifpaf_amd import synth
    from oracle import port
    seed, people, H, W, sr = case
    cif, caf = synth.synth_fields(seed, people, height=H, width=W, size_range=sr)
    fc = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0, nms_instance_threshold=0.0,
              nms_keypoint_threshold=0.0)
    for fields, kw in (((cif, caf), None), ((cif, caf), fc), ((to_bf16(cif), to_bf16(caf)), None)):
        params = port.default_params(**kw) if kw else None
        if kw:
            ref.apply_params(params)
        try:
            r_out, r_ids, r_hr = ref.decode(fields[0], 8, fields[1], 8, coco_skeleton0)
        finally:
            if kw:
                ref.reset_statics()
        o_out, o_ids, o_hr = port.decode(fields[0], 8, fields[1], 8, coco_skeleton0, params=params, return_cifhr=True)
        assert np.array_equal(r_hr, o_hr)
        assert r_out.shape == o_out.shape and len(o_out) >= 20 and np.array_equal(r_out, o_out)


def test_host_tracking_classes_run_the_reference_bookkeeping_on_our_annotations(monkeypatch):
    """``tracking.host_classes``: the reference's own TrackingPose / PoseSimilarity with a pose generator that returns THIS
    package's Annotation objects (here the oracle's decode, as in tests/test_tracking_pose.py) -- over the golden videos the
    reference's tracker produced with its own decoder; and their default pose generator is the HIP CifCaf."""
    from oracle import reference_python
    import os
    if not os.path.isdir(reference_python.REF_SRC):
        pytest.skip('reference sources not present')
    opp = reference_python.load()
    import sys
    import torch
    from common import TRACKING_VIDEOS
    from openpifpaf_amd import decoder, synth, tracking
    import test_tracking_pose as ttp
    host_tp, host_ps = tracking.host_classes(opp)
    assert tracking.host_classes(opp) == (host_tp, host_ps)                # (one pair of classes per package object)
    golden = np.load(ttp.GOLDEN)
    ref_track_annotation = sys.modules['openpifpaf.decoder.track_annotation'].TrackAnnotation
    for v, video in enumerate(TRACKING_VIDEOS):
        cif, caf, tcaf = ttp.metas()
        keypoints = list(cif.keypoints) * 2
        skeleton = list(caf.skeleton) + [(k + 1, k + 18) for k in range(17)]
        ref_track_annotation.track_id_counter = 0
        generator = ttp.OraclePoseGenerator(keypoints, skeleton)
        generator.occupancy_visualizer = None                               # (decoder.CifCaf has the attribute, see below)
        tracker = host_tp(cif, caf, tcaf, pose_generator=generator)
        seed, people, n_frames, appear = video
        for t, fields in enumerate(synth.synth_tracking_sequence(seed, people, n_frames, appear=appear)):
            anns = tracker([torch.from_numpy(f) for f in fields])
            assert [a.id_ for a in anns] == golden['video%d_frame%d_ids' % (v, t)].tolist(), 'video %d frame %d' % (v, t)
            got = np.asarray([a.data for a in anns], dtype=np.float32).reshape(-1, 17, 3)
            want = golden['video%d_frame%d_data' % (v, t)]
            assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4, 'video %d frame %d' % (v, t)
    # the default pose generator is this package's CifCaf, built on the TRACKING metas (34 keypoints, 19 + 17 bones); the class
    # itself needs a device, so a stand-in records what it is constructed with
    built = []

    class Recorder:
        occupancy_visualizer = None

        def __init__(self, cif_metas, caf_metas):
            built.append((len(cif_metas[0].keypoints), len(caf_metas[0].skeleton)))

    monkeypatch.setattr(tracking, 'CifCaf', Recorder)
    cif, caf, tcaf = ttp.metas()
    tp = host_tp(cif, caf, tcaf)
    ps = host_ps(cif, caf)
    assert type(tp.pose_generator) is Recorder and type(ps.pose_generator) is Recorder
    assert built == [(34, 19 + 17), (17, 19)]
    assert decoder.CifCaf.occupancy_visualizer is None                      # tracking_pose.py:160 reads it after the soft NMS
