"""The reference's tracking setup (decoder/tracking_pose.py:47-80,163-217): n_keypoints = 34 > 17 CIF fields,
single-frame skeleton + 17 temporal bones, previous-frame poses as initial annotations."""
import hashlib
import os

import numpy as np
import pytest

from common import TRACKING_CASES, compare_annotations, tracking_problem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tracking_golden.npz')


@pytest.fixture(scope='module')
def golden():
    return np.load(GOLDEN)


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize('i', range(len(TRACKING_CASES)))
def test_oracle_matches_reference_golden(golden, i):
    from oracle import port
    cif, caf, skel0, init, ids = tracking_problem(*TRACKING_CASES[i])
    assert _sha(cif, caf, init) == golden['case%d_input_sha256' % i].tobytes().decode()
    ann, got_ids = port.decode(cif, 8, caf, 8, skel0, n_keypoints=34, initial_annotations=init, initial_ids=ids)
    assert np.array_equal(ann, golden['case%d_annotations' % i]) and np.array_equal(got_ids, golden['case%d_ids' % i])
    ann0, _ = port.decode(cif, 8, caf, 8, skel0, n_keypoints=34)
    assert np.array_equal(ann0, golden['case%d_annotations_no_initial' % i])
    # previous-frame joints are reachable only through initial annotations (their CAF entries are rescored
    # against a CIF field that does not exist, caf_scored.cpp:15-18)
    assert not (ann0[:, 17:, 0] > 0).any() and (ann[: len(init), 17:, 0] > 0).any()
    # tracked poses keep their ids, new people get -1
    assert set(got_ids.tolist()) <= set(ids.tolist()) | {-1}


@pytest.mark.gpu
@pytest.mark.parametrize('i', range(len(TRACKING_CASES)))
def test_hip_path_equals_oracle_and_golden(golden, i):
    torch = pytest.importorskip('torch')
    from openpifpaf_amd import native
    cif, caf, skel0, init, ids = tracking_problem(*TRACKING_CASES[i])
    dec = native.CifCaf(34, torch.from_numpy(skel0))
    out, got_ids = dec.call_with_initial_annotations(
        torch.from_numpy(cif).cuda(), 8, torch.from_numpy(caf).cuda(), 8,
        torch.from_numpy(init).cuda(), torch.from_numpy(ids).cuda())
    want = golden['case%d_annotations' % i]
    assert out.shape == want.shape, (out.shape, want.shape)
    ok, msg = compare_annotations(out.cpu().numpy(), want)
    assert ok, msg
    assert np.array_equal(got_ids.cpu().numpy(), golden['case%d_ids' % i])
    out0, _ = dec.call(torch.from_numpy(cif).cuda(), 8, torch.from_numpy(caf).cuda(), 8)
    ok, msg = compare_annotations(out0.cpu().numpy(), golden['case%d_annotations_no_initial' % i])
    assert ok, msg
    # batched: both problems in one launch, initial annotations per image
    cifs = torch.from_numpy(np.stack([cif, cif])).cuda()
    cafs = torch.from_numpy(np.stack([caf, caf])).cuda()
    inits = torch.from_numpy(np.stack([init, np.zeros_like(init)])).cuda()
    idss = torch.from_numpy(np.stack([ids, ids])).cuda()
    bo, bi, bc = dec.call_batch(cifs, 8, cafs, 8, inits, idss)
    n0 = int(bc[0])
    ok, msg = compare_annotations(bo[0, :n0].cpu().numpy(), want)
    assert ok, msg


@pytest.mark.gpu
def test_torchscript_binding_tracking_setup():
    torch = pytest.importorskip('torch')
    from openpifpaf_amd import torchscript
    cif, caf, skel0, init, ids = tracking_problem(*TRACKING_CASES[0])
    dec = torchscript.load().CifCaf(34, torch.from_numpy(skel0))
    out, got_ids = dec.call_with_initial_annotations(torch.from_numpy(cif), 8, torch.from_numpy(caf), 8,
                                                     torch.from_numpy(init), torch.from_numpy(ids))
    want = np.load(GOLDEN)['case0_annotations']
    ok, msg = compare_annotations(out.numpy(), want)
    assert ok and not out.is_cuda, msg
