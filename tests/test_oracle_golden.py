"""Pins the CPU oracle (oracle/cifcaf_oracle.cpp) against the committed golden
vectors, which are outputs of the REAL reference decoder (tests/golden/make_golden.py).
The reference's own tests hold no known-answer vectors for this path."""
import hashlib
import os

import numpy as np
import pytest

from common import GOLDEN_CASES

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(HERE, 'golden', 'cifcaf_golden.npz'))


@pytest.fixture(scope='module')
def port():
    from oracle import port as p
    p.lib()
    return p


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def text(a):
    return bytes(a).decode()


def case_fields(i):
    from openpifpaf_amd import synth
    seed, people, H, W = GOLDEN_CASES[i]
    return synth.synth_fields(seed, people, height=H, width=W)


@pytest.mark.parametrize('i', range(len(GOLDEN_CASES)))
def test_inputs_regenerate_bit_identically(golden, i):
    cif, caf = case_fields(i)
    assert digest(cif, caf) == text(golden['case%d_input_sha256' % i]), \
        'the field synthesiser no longer reproduces the golden inputs (numpy RNG / generator change)'


@pytest.mark.parametrize('i', range(len(GOLDEN_CASES)))
def test_decode_matches_reference_golden(golden, port, coco_skeleton0, i):
    cif, caf = case_fields(i)
    ann, ids, hr = port.decode(cif, 8, caf, 8, coco_skeleton0, return_cifhr=True)
    want = golden['case%d_annotations' % i]
    assert ann.shape == want.shape
    assert np.array_equal(ann, want), 'max |delta| %g' % np.abs(ann - want).max()
    assert np.array_equal(ids, golden['case%d_ids' % i])
    assert digest(hr) == text(golden['case%d_cifhr_sha256' % i])
    assert len(want) > 0, 'golden case without poses pins nothing'


@pytest.mark.parametrize('i', range(len(GOLDEN_CASES)))
def test_stages_match_reference_golden(golden, port, coco_skeleton0, i):
    cif, caf = case_fields(i)
    hr = port.cifhr_accumulate(cif, 8)
    f, v = port.cifseeds(cif, 8, hr)
    assert np.array_equal(f, golden['case%d_seed_f' % i])
    assert np.array_equal(v, golden['case%d_seed_vxys' % i])
    fwd, bwd = port.cafscored(caf, 8, hr, cif.shape, 8, coco_skeleton0)
    counts = np.array([[len(a), len(b)] for a, b in zip(fwd, bwd)], dtype=np.int32)
    assert np.array_equal(counts, golden['case%d_caf_counts' % i])
    assert digest(*fwd, *bwd) == text(golden['case%d_caf_sha256' % i])


@pytest.mark.parametrize('i', range(len(GOLDEN_CASES)))
def test_force_complete_matches_reference_golden(golden, port, coco_skeleton0, i):
    cif, caf = case_fields(i)
    p = port.default_params(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                            nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
    ann, _ = port.decode(cif, 8, caf, 8, coco_skeleton0, params=p)
    want = golden['case%d_annotations_fc' % i]
    assert ann.shape == want.shape and np.array_equal(ann, want)


def test_grow_connection_blend_known_answers(golden, port):
    rows = golden['blend_rows']
    for q, want in zip(golden['blend_queries'], golden['blend_answers']):
        got = port.grow_connection_blend(rows, q[0], q[1], q[2], q[3], bool(q[4]))
        assert np.array_equal(got, want), (q, got, want)


def test_edge_cases(port, coco_skeleton0):
    # empty fields: nothing passes any threshold
    cif = np.zeros((17, 5, 9, 9), dtype=np.float32)
    caf = np.zeros((19, 8, 9, 9), dtype=np.float32)
    ann, ids = port.decode(cif, 8, caf, 8, coco_skeleton0)
    assert ann.shape == (0, 17, 4) and ids.shape == (0,)
    # a single confident cell: one seed, a one-joint pose, removed by the instance threshold
    cif[3, 1, 4, 4] = 0.9
    cif[3, 2, 4, 4] = 4.0
    cif[3, 3, 4, 4] = 4.0
    cif[3, 4, 4, 4] = 1.0
    hr = port.cifhr_accumulate(cif, 8)
    assert hr.max() > 1.0 and hr[3].max() == hr.max() and (hr[:3] == 0).all()
    f, v = port.cifseeds(cif, 8, hr, params=port.default_params(seed_threshold=0.05))
    assert f.tolist() == [3] and v[0, 1] == 32.0 and v[0, 2] == 32.0
    ann, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
    assert len(ann) == 0
    ann, _ = port.decode(cif, 8, caf, 8, coco_skeleton0,
                         params=port.default_params(seed_threshold=0.05, nms_instance_threshold=0.0,
                                                    nms_keypoint_threshold=0.0))
    assert len(ann) == 1 and (ann[0, :, 0] > 0).sum() == 1
    # cells on the border: regressions outside the map are clamped, never out of bounds
    cif[5, 1, 0, 0] = 0.95
    cif[5, 2, 0, 0] = -3.0
    cif[5, 3, 0, 0] = 40.0
    cif[5, 4, 0, 0] = 2.0
    port.decode(cif, 8, caf, 8, coco_skeleton0)
