"""CifDet: oracle vs the reference's golden outputs (CPU), HIP path vs
oracle (GPU), host-side NMS."""
import hashlib
import os

import numpy as np
import pytest

from common import DET_CASES

HERE = os.path.dirname(os.path.abspath(__file__))


def field_of(i):
    from openpifpaf_amd import synth
    seed, n, H, W = DET_CASES[i]
    return synth.synth_det_field(seed, n, height=H, width=W)


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(HERE, 'golden', 'cifcaf_golden.npz'))


@pytest.mark.parametrize('i', range(len(DET_CASES)))
def test_oracle_matches_reference_golden(golden, i):
    from oracle import port
    field = field_of(i)
    assert hashlib.sha256(field.tobytes()).hexdigest() == bytes(golden['det%d_input_sha256' % i]).decode()
    cat, sc, bx = port.cifdet_decode(field, 8)
    assert np.array_equal(cat, golden['det%d_categories' % i])
    assert np.array_equal(sc, golden['det%d_scores' % i])
    assert np.array_equal(bx, golden['det%d_boxes' % i])
    assert len(cat) >= 1 and len(cat) <= 120          # max_detections_before_nms, cifdet.cpp:16,66


def test_nms_and_annotation_det():
    from openpifpaf_amd.decoder import _nms_keep
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], dtype=np.float64)
    scores = np.array([0.9, 0.8, 0.7, 0.95])
    keep = _nms_keep(boxes, scores, 0.5)
    assert keep.tolist() == [3, 2]
    from openpifpaf_amd.annotation import AnnotationDet
    ann = AnnotationDet(['a', 'b']).set(2, 0.71234, [1.0, 2.0, 3.0, 4.0])
    assert ann.json_data() == {'category_id': 2, 'category': 'b', 'score': 0.712, 'bbox': [1.0, 2.0, 3.0, 4.0]}
    inv = ann.inverse_transform({'offset': (-1.0, -2.0), 'scale': (2.0, 2.0), 'hflip': False})
    assert np.allclose(inv.bbox, [0.0, 0.0, 1.5, 2.0])


@pytest.mark.gpu
@pytest.mark.parametrize('i', range(len(DET_CASES)))
def test_hip_cifdet_matches_oracle(i):
    import torch
    from openpifpaf_amd import native
    from oracle import port
    field = field_of(i)
    want = port.cifdet_decode(field, 8)
    det = native.CifDet()
    cat, sc, bx = det.call(torch.from_numpy(field).cuda(), 8)
    assert np.array_equal(cat.cpu().numpy(), want[0])
    assert np.array_equal(sc.cpu().numpy(), want[1])          # bit-exact: CifHr map and seeds are
    assert np.array_equal(bx.cpu().numpy(), want[2])


@pytest.mark.gpu
def test_hip_cifdet_batch_and_decoder_class():
    import torch
    from openpifpaf_amd import decoder, headmeta, native, synth
    from oracle import port
    fields = np.stack([synth.synth_det_field(50 + b, 3 + 5 * b) for b in range(6)])
    det = native.CifDet()
    cat, sc, bx, cnt = det.call_batch(torch.from_numpy(fields).cuda(), 8)
    for b in range(6):
        want = port.cifdet_decode(fields[b], 8)
        n = int(cnt[b])
        assert n == len(want[0])
        assert np.array_equal(cat[b, :n].cpu().numpy(), want[0])
        assert np.array_equal(sc[b, :n].cpu().numpy(), want[1])
        assert np.array_equal(bx[b, :n].cpu().numpy(), want[2])
    meta = headmeta.CifDet('cifdet', 'synthetic', categories=['c%d' % i for i in range(8)])
    meta.head_index, meta.base_stride, meta.upsample_stride = 0, 16, 2
    dec = decoder.CifDet.factory([meta])[0]
    anns = dec([torch.from_numpy(fields[2]).cuda()])
    assert len(anns) >= 1 and all(a.score > dec.instance_threshold for a in anns)
    assert all(a.bbox[2] > 0 and a.bbox[3] > 0 for a in anns)
