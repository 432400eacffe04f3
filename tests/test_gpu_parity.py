"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle
on identical, seeded field tensors.  Integer/index results bit-exact; the CifHr map
and the CAF lists bit-exact (ordered accumulation, no FMA); keypoints within 1e-4."""
import numpy as np
import pytest

from common import TOL, compare_annotations

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def native():
    from openpifpaf_amd import native as n
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return n


@pytest.fixture(scope='module')
def port():
    from oracle import port as p
    return p


def fields(seed, people, H=81, W=81):
    from openpifpaf_amd import synth
    return synth.synth_fields(seed, people, height=H, width=W)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CASES = [(0, 1, 41, 41), (1, 5, 81, 81), (2, 10, 81, 81), (3, 20, 81, 81), (4, 3, 33, 49), (5, 6, 49, 33)]


@pytest.mark.parametrize('seed,people,H,W', CASES)
def test_cifhr_bit_exact(native, port, seed, people, H, W):
    cif, _ = fields(seed, people, H, W)
    ref = port.cifhr_accumulate(cif, 8)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    got, rev = hr.get_accumulated()
    assert rev == 1.0
    got = got.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), 'max |delta| %g, %d cells differ' % (
        np.abs(got - ref).max(), (got != ref).sum())


def test_cifhr_min_scale_and_factor(native, port):
    cif, _ = fields(11, 5)
    ref = port.cifhr_accumulate(cif, 8, min_scale=6.0, factor=0.5)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8, 6.0, 0.5)
    assert np.array_equal(hr.get_accumulated()[0].cpu().numpy(), ref)


@pytest.mark.parametrize('seed,people,H,W', CASES)
def test_cifseeds_exact(native, port, seed, people, H, W):
    cif, _ = fields(seed, people, H, W)
    ref_hr = port.cifhr_accumulate(cif, 8)
    ref_f, ref_v = port.cifseeds(cif, 8, ref_hr)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    seeds = native.CifSeeds(hr)
    seeds.fill(dev(cif), 8)
    f, v = seeds.get()
    f, v = f.cpu().numpy(), v.cpu().numpy()
    assert len(f) == len(ref_f)
    assert np.all(np.diff(v[:, 0]) <= 0), 'not sorted by score'
    if len(np.unique(ref_v[:, 0])) == len(ref_v):      # no score ties: order is fully determined
        assert np.array_equal(f, ref_f)
        assert np.array_equal(v, ref_v)
    else:                                             # compare as sets
        a = sorted(map(tuple, np.column_stack([f, v]).tolist()))
        b = sorted(map(tuple, np.column_stack([ref_f, ref_v]).tolist()))
        assert a == b


@pytest.mark.parametrize('seed,people,H,W', CASES)
def test_cafscored_exact(native, port, coco_skeleton0, seed, people, H, W):
    cif, caf = fields(seed, people, H, W)
    ref_hr = port.cifhr_accumulate(cif, 8)
    ref_f, ref_b = port.cafscored(caf, 8, ref_hr, cif.shape, 8, coco_skeleton0)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    cs = native.CafScored(hr, cif.shape, 8)
    cs.fill(dev(caf), 8, torch.from_numpy(coco_skeleton0))
    fwd, bwd = cs.get()
    for a in range(len(ref_f)):
        assert np.array_equal(fwd[a].cpu().numpy(), ref_f[a]), 'forward list %d' % a
        assert np.array_equal(bwd[a].cpu().numpy(), ref_b[a]), 'backward list %d' % a


def test_cafscored_force_complete_threshold(native, port, coco_skeleton0):
    cif, caf = fields(7, 5)
    ref_hr = port.cifhr_accumulate(cif, 8)
    ref_f, ref_b = port.cafscored(caf, 8, ref_hr, cif.shape, 8, coco_skeleton0, score_th=0.001)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    cs = native.CafScored(hr, cif.shape, 8, 0.001, 0.1)
    cs.fill(dev(caf), 8, torch.from_numpy(coco_skeleton0))
    fwd, bwd = cs.get()
    for a in range(len(ref_f)):
        assert np.array_equal(fwd[a].cpu().numpy(), ref_f[a])
        assert np.array_equal(bwd[a].cpu().numpy(), ref_b[a])


def test_grow_connection_blend(native, port, coco_skeleton0):
    cif, caf = fields(2, 10)
    ref_hr = port.cifhr_accumulate(cif, 8)
    ref_f, _ = port.cafscored(caf, 8, ref_hr, cif.shape, 8, coco_skeleton0)
    rng = np.random.default_rng(0)
    checked = 0
    for rows in ref_f[:8]:
        if len(rows) == 0:
            continue
        rows_d = dev(rows)
        for _ in range(6):
            r = rows[rng.integers(len(rows))]
            x, y = float(r[1] + rng.normal(0, 1.0)), float(r[2] + rng.normal(0, 1.0))
            s = float(rng.uniform(2.0, 30.0))
            for only_max in (False, True):
                for fs in (1.0, 4.0):
                    want = port.grow_connection_blend(rows, x, y, s, fs, only_max)
                    got = np.asarray(native.grow_connection_blend(rows_d, x, y, s, fs, only_max))
                    assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (got, want)
                    checked += 1
    assert checked > 50
    # empty list and far-away query -> all-zero joint
    assert native.grow_connection_blend(torch.zeros((0, 7)).cuda(), 1.0, 2.0, 3.0) == [0.0, 0.0, 0.0, 0.0]
    assert native.grow_connection_blend(dev(ref_f[0]), -1e4, -1e4, 1.0) == [0.0, 0.0, 0.0, 0.0]


def test_grow_connection_blend_nan(native, port):
    """Garbage input: list entries with a NaN coordinate or confidence.  The reference lets a NaN coordinate through its
    window test (cifcaf.cpp:54-57: `if (x1 < lo) continue` is false for a NaN), but the entry's score is then NaN and never
    satisfies `>=` / `>` (:65-73): it is ignored.  The kernel skips it at the window test.  Same joint, for every mix."""
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(3, 200))
        rows = np.zeros((n, 7), dtype=np.float32)
        rows[:, 0] = rng.uniform(0.3, 1.0, n)
        rows[:, 1] = 100.0 + rng.normal(0, 2.5, n)
        rows[:, 2] = 100.0 + rng.normal(0, 2.5, n)
        rows[:, 3], rows[:, 4] = rng.uniform(0, 600, n), rng.uniform(0, 600, n)
        rows[:, 5], rows[:, 6] = 4.0, rng.uniform(1, 8, n)
        bad = rows.copy()
        k = rng.integers(0, n, size=max(1, n // 4))
        bad[k, rng.integers(0, 3, size=len(k))] = np.nan             # confidence, x1 or y1
        for only_max in (False, True):
            want = port.grow_connection_blend(bad, 100.0, 100.0, 6.0, 1.0, only_max)
            got = np.asarray(native.grow_connection_blend(dev(bad), 100.0, 100.0, 6.0, 1.0, only_max))
            assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (trial, got, want)
            keep = np.ones(n, dtype=bool)
            keep[k] = False
            assert np.array_equal(want, port.grow_connection_blend(rows[keep], 100.0, 100.0, 6.0, 1.0, only_max))


def test_grow_connection_blend_window_edges(native, port):
    """Entries exactly on, one float below and one float above the four edges of the filter window
    (cifcaf.cpp:54-57 compares the float entry with double bounds): the kernel tests floats against directed-rounded
    float bounds, which must select exactly the same entries -- for queries whose bounds are and are not floats."""
    rng = np.random.default_rng(7)
    checked = 0
    for _ in range(40):
        x, y = float(rng.uniform(5, 600)), float(rng.uniform(5, 600))
        if rng.random() < 0.5:
            x, y = float(np.float32(x)), float(np.float32(y))           # bounds representable as floats
        s = float(rng.choice([1.0, 3.0, 7.3, 12.0, 25.5]))
        for fs in (1.0, 4.0):
            sigma = float(np.float32(fs * max(s, 0.5) / 2.0))               # :47 (a float in the reference)
            rows = []
            for bound, other in ((x - sigma, y), (x + sigma, y)):
                f = np.float32(bound)
                for v in (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))):
                    rows.append([0.5 + 0.4 * rng.random(), float(v), other, 100.0 + len(rows), 200.0, 4.0, 4.0])
            for bound, other in ((y - sigma, x), (y + sigma, x)):
                f = np.float32(bound)
                for v in (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))):
                    rows.append([0.5 + 0.4 * rng.random(), other, float(v), 100.0 + len(rows), 200.0, 4.0, 4.0])
            rows = np.asarray(rows, dtype=np.float32)
            for only_max in (False, True):
                want = port.grow_connection_blend(rows, x, y, s, fs, only_max)
                got = np.asarray(native.grow_connection_blend(dev(rows), x, y, s, fs, only_max))
                assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (x, y, s, fs, got, want)
                checked += 1
    assert checked == 160


def test_grow_connection_blend_ties(native, port):
    """Exactly equal scores: the reference's '>=' / '>' rules pick by list position."""
    base = np.array([[0.8, 10.0, 10.0, 50.0, 60.0, 4.0, 4.0]], dtype=np.float32)
    for n in (2, 3, 5, 130, 700):
        rows = np.repeat(base, n, axis=0)
        rows[:, 3] += np.arange(n, dtype=np.float32) * 0.25       # distinguishable targets
        for x in (10.0, 10.5):
            want = port.grow_connection_blend(rows, x, 10.0, 8.0)
            got = np.asarray(native.grow_connection_blend(dev(rows), x, 10.0, 8.0))
            assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (n, got, want)


@pytest.mark.parametrize('seed,people,H,W', CASES + [(6, 15, 81, 81), (7, 8, 81, 81), (8, 2, 81, 81)])
def test_decode_parity(native, port, coco_skeleton0, seed, people, H, W):
    cif, caf = fields(seed, people, H, W)
    want, want_ids = port.decode(cif, 8, caf, 8, coco_skeleton0)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, ids = dec.call(dev(cif), 8, dev(caf), 8)
    ok, msg = compare_annotations(got.cpu().numpy(), want)
    assert ok, msg
    assert np.array_equal(ids.cpu().numpy(), want_ids)
    hr, rev = dec.get_cifhr()
    assert rev == 1.0 and tuple(hr.shape) == (17, (H - 1) * 8 + 1, (W - 1) * 8 + 1)


def test_decode_batch_matches_single(native, port, coco_skeleton0):
    from openpifpaf_amd import synth
    cifs, cafs = synth.synth_batch(8, seed0=100)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    for b in range(8):
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        assert counts[b] == len(want)
        ok, msg = compare_annotations(out[b, :counts[b]], want)
        assert ok, 'image %d: %s' % (b, msg)


def test_decode_force_complete(native, port, coco_skeleton0):
    from openpifpaf_amd import _lib
    for seed, people in [(20, 3), (21, 8), (22, 15)]:
        cif, caf = fields(seed, people)
        kw = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                  nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0, params=port.default_params(**kw))
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=512)
        out, ids, counts = dec.call_batch(dev(cif)[None], 8, dev(caf)[None], 8, params=_lib.default_params(**kw))
        n = int(counts[0])
        assert n == len(want)
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert ok, 'seed %d: %s' % (seed, msg)
        assert (want[:, :, 0] > 0).all(), 'force complete leaves no joint empty'


def test_decode_initial_annotations(native, port, coco_skeleton0):
    cif, caf = fields(30, 4)
    first, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
    assert len(first) >= 2
    init = first[:2].copy()
    init[:, 5:, :] = 0.0                      # keep only the head joints: the decoder must regrow the rest
    init_ids = np.array([7, 9], dtype=np.int64)
    want, want_ids = port.decode(cif, 8, caf, 8, coco_skeleton0, initial_annotations=init, initial_ids=init_ids)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, ids = dec.call_with_initial_annotations(dev(cif), 8, dev(caf), 8, dev(init), dev(init_ids))
    ok, msg = compare_annotations(got.cpu().numpy(), want)
    assert ok, msg
    assert np.array_equal(ids.cpu().numpy(), want_ids)
    assert {7, 9} <= set(want_ids.tolist())


def test_decode_greedy_and_no_reverse_match(native, port, coco_skeleton0):
    from openpifpaf_amd import _lib
    cif, caf = fields(40, 6)
    for kw in (dict(greedy=1), dict(reverse_match=0), dict(keypoint_threshold=0.3, keypoint_threshold_rel=0.8)):
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0, params=port.default_params(**kw))
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
        out, ids, counts = dec.call_batch(dev(cif)[None], 8, dev(caf)[None], 8, params=_lib.default_params(**kw))
        n = int(counts[0])
        assert n == len(want), kw
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert ok, '%s: %s' % (kw, msg)


def test_decode_empty_fields(native, coco_skeleton0):
    cif = np.zeros((17, 5, 41, 41), dtype=np.float32)
    caf = np.zeros((19, 8, 41, 41), dtype=np.float32)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, ids = dec.call(dev(cif), 8, dev(caf), 8)
    assert got.shape == (0, 17, 4) and ids.shape == (0,)


def test_decode_all_active_adversarial(native, port, coco_skeleton0):
    """Random-initialised-network statistics: every cell is active.  With the seed
    rescoring ablated (raw confidences are the seed scores) 31x31 cells x 17 fields
    give ~15k seeds > 8192: exercises the LDS-blocked bitonic sort and long lists."""
    from openpifpaf_amd import _lib, synth
    cif, caf = synth.adversarial_fields(3, height=31, width=31)
    kw = dict(ablation_cifseeds_no_rescore=1)
    ref_hr = port.cifhr_accumulate(cif, 8)
    ref_f, ref_v = port.cifseeds(cif, 8, ref_hr, params=port.default_params(**kw))
    assert len(ref_f) > 8192
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    assert np.array_equal(hr.get_accumulated()[0].cpu().numpy(), ref_hr)
    seeds = native.CifSeeds(hr)
    seeds.fill(dev(cif), 8, params=_lib.default_params(**kw))
    f, v = seeds.get()
    f, v = f.cpu().numpy(), v.cpu().numpy()
    assert len(f) == len(ref_f)
    assert np.all(np.diff(v[:, 0]) <= 0)
    if len(np.unique(ref_v[:, 0])) == len(ref_v):
        assert np.array_equal(f, ref_f) and np.array_equal(v, ref_v)
    else:
        assert np.array_equal(np.sort(v[:, 0]), np.sort(ref_v[:, 0]))
    for kw in (dict(), dict(ablation_cifseeds_no_rescore=1)):
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0, params=port.default_params(**kw))
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=512)
        out, ids, counts = dec.call_batch(dev(cif)[None], 8, dev(caf)[None], 8, params=_lib.default_params(**kw))
        n = int(counts[0])
        assert n == len(want), kw
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert ok, '%s: %s' % (kw, msg)


@pytest.mark.parametrize('seed,people,H,W', [(0, 2, 41, 41), (1, 4, 81, 81)])
def test_decode_wholebody_133_keypoints(native, port, seed, people, H, W):
    """BASELINE config 4 shapes: 133 keypoints, 160 bones (max joint degree 6, K > one wave)."""
    from openpifpaf_amd import constants, synth
    wb = constants.wholebody()
    skel0 = np.asarray(wb['skeleton'], dtype=np.int64) - 1
    cif, caf = synth.synth_fields(seed, people, height=H, width=W, pose=wb['standing_pose'],
                                  skeleton=wb['skeleton'], size_range=(0.6, 0.95))
    assert cif.shape == (133, 5, H, W) and caf.shape == (160, 8, H, W)
    want, _ = port.decode(cif, 8, caf, 8, skel0)
    assert len(want) >= 1
    dec = native.CifCaf(133, torch.from_numpy(skel0))
    got, ids = dec.call(dev(cif), 8, dev(caf), 8)
    ok, msg = compare_annotations(got.cpu().numpy(), want)
    assert ok, msg
    # stage parity at this shape too
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    assert np.array_equal(hr.get_accumulated()[0].cpu().numpy(), port.cifhr_accumulate(cif, 8))
    # force complete on the big skeleton (flood fill over 160 bones, frontier ties everywhere)
    from openpifpaf_amd import _lib
    kw = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
              nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
    want_fc, _ = port.decode(cif, 8, caf, 8, skel0, params=port.default_params(**kw))
    dec = native.CifCaf(133, torch.from_numpy(skel0), max_annotations=256)
    out, _, counts = dec.call_batch(dev(cif)[None], 8, dev(caf)[None], 8, params=_lib.default_params(**kw))
    n = int(counts[0])
    assert n == len(want_fc)
    ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want_fc)
    assert ok, 'force complete: ' + msg


def test_static_getset_roundtrip(native):
    C = native.CifCaf
    old = C.get_keypoint_threshold()
    try:
        C.set_keypoint_threshold(0.25)
        assert C.get_keypoint_threshold() == 0.25
        C.set_force_complete(True)
        assert C.get_force_complete() is True
    finally:
        C.set_keypoint_threshold(old)
        C.set_force_complete(False)
    assert native.CifHr.get_threshold() == 0.3 and native.CifSeeds.get_threshold() == 0.2
    assert native.CafScored.get_default_score_th() == 0.3 and native.NMSKeypoints.get_suppression() == 1e-5


def test_full_size_batch_properties(native, port, coco_skeleton0):
    """BASELINE config sizes (batch 32, 17x81x81 + 19x81x81): properties that do not need the
    oracle on every image -- determinism, batch-order equivariance, sortedness, bounds --
    plus an oracle comparison on a sample of the batch."""
    from openpifpaf_amd import synth
    B = 32
    cifs, cafs = synth.synth_batch(B, seed0=500)
    cif_d, caf_d = dev(cifs), dev(cafs)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out1, ids1, cnt1 = [t.clone() for t in dec.call_batch(cif_d, 8, caf_d, 8)]
    # seeds of the same call: sorted by score, scores within [seed_th, 1], coordinates inside the map
    n_seeds = dec.workspace_view('seed_count', torch.int32)[:B].cpu().numpy()
    vxys = dec.workspace_view('seed_vxys', torch.float32).view(B, -1, 4)
    for b in range(B):
        v = vxys[b, :n_seeds[b]].cpu().numpy()
        assert np.all(np.diff(v[:, 0]) <= 0) and v[:, 0].min() >= 0.2 and v[:, 0].max() <= 1.0
    # the CifHr map holds 0 (untouched) or 1 + value with value in (0, 1]
    hr, rev = dec.get_cifhr(image=3)
    hr = hr.cpu().numpy()
    assert rev == 1.0 and hr.shape == (17, 641, 641)
    touched = hr[hr != 0]
    assert touched.min() > 1.0 and touched.max() <= 2.0
    # idempotence / determinism: same call, same bits
    out2, ids2, cnt2 = dec.call_batch(cif_d, 8, caf_d, 8)
    assert torch.equal(cnt1, cnt2)
    for b in range(B):
        assert torch.equal(out1[b, :int(cnt1[b])], out2[b, :int(cnt2[b])])
    # batch-order equivariance: images are independent units
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    out3, ids3, cnt3 = dec.call_batch(cif_d[perm], 8, caf_d[perm], 8)
    assert torch.equal(cnt3, cnt1[perm])
    for b in range(B):
        n = int(cnt3[b])
        assert torch.equal(out3[b, :n], out1[perm[b], :n])
    # outputs: scores sorted (mean of v), keypoints inside the padded image
    o = out1.cpu().numpy()
    c = cnt1.cpu().numpy()
    assert c.sum() > 150 and c.max() <= dec.max_annotations
    for b in range(B):
        a = o[b, :c[b]]
        if len(a):
            assert np.all(np.diff(a[:, :, 0].mean(axis=1)) <= 1e-7)
            present = a[:, :, 0] > 0
            assert a[:, :, 1][present].min() > -64 and a[:, :, 1][present].max() < 704
    # oracle on a sample
    for b in (0, 3, 7, 19, 31):
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        ok, msg = compare_annotations(o[b, :c[b]], want)
        assert ok, 'image %d: %s' % (b, msg)


def test_repeatability_and_wide_oracle_sweep(native, port, coco_skeleton0):
    """Race screen for the speculative multi-wave association kernel: 48 different images, every
    one against the oracle, and 10 repeated launches must return identical bits."""
    from openpifpaf_amd import synth
    B = 48
    people = (1, 2, 4, 6, 9, 12, 16, 24)
    cifs, cafs = synth.synth_batch(B, seed0=900, people=people)
    cif_d, caf_d = dev(cifs), dev(cafs)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    first = None
    for rep in range(10):
        out, ids, cnt = dec.call_batch(cif_d, 8, caf_d, 8)
        cur = (out.clone(), cnt.clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[1], first[1]), 'repeat %d: counts differ' % rep
            for b in range(B):
                n = int(cur[1][b])
                assert torch.equal(cur[0][b, :n], first[0][b, :n]), 'repeat %d image %d differs' % (rep, b)
    o, c = first[0].cpu().numpy(), first[1].cpu().numpy()
    for b in range(B):
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        ok, msg = compare_annotations(o[b, :c[b]], want)
        assert ok, 'image %d (%d people): %s' % (b, people[b % len(people)], msg)


def test_single_image_repeat_stress(native, port, coco_skeleton0):
    """Regression: with plain workgroup barriers the association kernel's waves could disagree on an
    occupancy test (global stores still in flight) -- about one run in four of THIS input came back
    with joints wrongly suppressed.  150 decodes through one handle must all equal the oracle."""
    from openpifpaf_amd import synth
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    for seed, people, size in ((32, 6, 65), (35, 9, 65)):
        cif, caf = synth.synth_fields(seed, people, height=size, width=size)
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
        cif_d, caf_d = dev(cif[None]), dev(caf[None])
        first = None
        for rep in range(150):
            out, ids, cnt = dec.call_batch(cif_d, 8, caf_d, 8)
            got = out[0, :int(cnt[0])].clone()
            if first is None:
                first = got
                ok, msg = compare_annotations(got.cpu().numpy(), want)
                assert ok, msg
            else:
                assert got.shape == first.shape and torch.equal(got, first), 'repeat %d differs' % rep


@pytest.mark.parametrize('cif_stride,caf_stride', [(12, 12), (4, 4), (16, 16)])
def test_decode_other_strides(native, port, coco_skeleton0, cif_stride, caf_stride):
    """Non-power-of-two strides make every `value * stride` inexact: exercises the float/double
    promotion order of every stage (the synthetic fields are in field units, so they decode at any stride)."""
    cif, caf = fields(60 + cif_stride, 4, 41, 41)
    ref_hr = port.cifhr_accumulate(cif, cif_stride)
    hr = native.CifHr()
    hr.accumulate(dev(cif), cif_stride)
    assert np.array_equal(hr.get_accumulated()[0].cpu().numpy(), ref_hr)
    want, _ = port.decode(cif, cif_stride, caf, caf_stride, coco_skeleton0)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, _ = dec.call(dev(cif), cif_stride, dev(caf), caf_stride)
    ok, msg = compare_annotations(got.cpu().numpy(), want)
    assert ok, msg
    assert len(want) >= 1


def test_decode_different_cif_and_caf_geometry(native, port, coco_skeleton0):
    """CAF head at half the CIF resolution (stride 16 vs 8): the reference allows it
    (cifcaf.cpp:140-157 takes the two strides separately)."""
    from openpifpaf_amd import synth
    cif, _ = synth.synth_fields(77, 3, height=41, width=41, size_range=(0.6, 0.95))
    _, caf16 = synth.synth_fields(77, 3, height=21, width=21, size_range=(0.6, 0.95))
    want, _ = port.decode(cif, 8, caf16, 16, coco_skeleton0)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, _ = dec.call(dev(cif), 8, dev(caf16), 16)
    ok, msg = compare_annotations(got.cpu().numpy(), want)
    assert ok, msg


def test_workspace_state_across_different_images(native, port, coco_skeleton0):
    """The CifHr map is cleared lazily per tile (clean-tile flags live in the workspace): a sequence of
    DIFFERENT images through one decoder -- crowded, sparse, empty, crowded again -- must give the oracle's
    map and poses every time; a workspace that was scribbled over recovers once its header is wiped."""
    from openpifpaf_amd import synth
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    seq = [(61, 14), (62, 1), (63, 0), (64, 9), (62, 1), (61, 14)]
    for step, (seed, people) in enumerate(seq + seq[:2]):
        cif, caf = synth.synth_fields(seed, people, height=49, width=57)
        if people == 0:
            cif[:, 1] = 0.0
            caf[:, 1] = 0.0
        if step == len(seq):                        # someone else used the memory: garbage everywhere, header wiped
            shape, ws = dec._last
            ws.view(torch.float32)[64:].uniform_(0.5, 3.0)
            ws[:256] = 0
        out, ids, cnt = dec.call_batch(dev(cif[None]), 8, dev(caf[None]), 8)
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
        want_hr = port.cifhr_accumulate(cif, 8)
        n = int(cnt[0])
        assert n == len(want), 'step %d: %d poses, oracle %d' % (step, n, len(want))
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert ok, 'step %d: %s' % (step, msg)
        hr, rev = dec.get_cifhr()
        got_hr = hr.cpu().numpy()
        assert got_hr.shape == want_hr.shape
        assert np.array_equal(got_hr, want_hr), 'step %d: stale or missing tiles (%d cells differ)' % (
            step, (got_hr != want_hr).sum())


def test_annotation_capacity_overflow_is_flagged(native, port, coco_skeleton0):
    """Too small a capacity: the overflow bit is raised, the count says how many rows are VALID, and those rows
    are real poses -- the first ones the sequential loop creates, after keypoint NMS among themselves
    (include/openpifpaf_amd.h: out_count_dev).  Rows behind them are never handed out as poses."""
    cif, caf = fields(91, 9)
    want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
    assert len(want) >= 6
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=3)
    out, ids, cnt = dec.call_batch(dev(cif[None]), 8, dev(caf[None]), 8)
    c = int(cnt[0])
    assert native.count_overflowed(c) and 1 <= native.count_rows(c) <= 3
    assert int(dec.workspace_view('status', torch.int32)[0]) >= len(want) - 3     # poses dropped
    rows = out[0, :native.count_rows(c)].cpu().numpy()
    for row in rows:                                 # each valid row is one of the real poses (NMS only lowers v)
        k0 = int(np.argmax(row[:, 0] > 0))
        match = [w for w in want if w[k0, 0] > 0 and np.abs(w[k0, 1:] - row[k0, 1:]).max() <= TOL]
        assert match, 'row is not a pose of the full decode'
        both = (row[:, 0] > 0) & (match[0][:, 0] > 0)
        assert np.abs(row[both] - match[0][both]).max() <= TOL
    from openpifpaf_amd import distributed
    assert len(distributed.unpack(out, ids, cnt)[0][0]) == native.count_rows(c)
    with pytest.raises(Exception, match='capacity overflow'):
        dec.call(dev(cif), 8, dev(caf), 8)
    # enough room again: same decoder class, full result
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=len(want))
    out, ids, cnt = dec.call_batch(dev(cif[None]), 8, dev(caf[None]), 8)
    ok, msg = compare_annotations(out[0, :int(cnt[0])].cpu().numpy(), want)
    assert int(cnt[0]) == len(want) and ok, msg


def test_random_geometry_sweep(native, port, coco_skeleton0):
    """Many small, odd-shaped fields with random people counts, sizes and noise: every one must equal the
    oracle (pool refill, tile bitmaps at plane borders, clamped boxes, single-row / single-column maps)."""
    from openpifpaf_amd import synth
    rng = np.random.default_rng(2024)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    for case in range(40):
        H, W = int(rng.integers(5, 40)), int(rng.integers(5, 40))
        people = int(rng.integers(0, 7))
        lo = float(rng.uniform(0.2, 0.6))
        cif, caf = synth.synth_fields(1000 + case, people, height=H, width=W,
                                      noise=float(rng.uniform(0.0, 0.3)), size_range=(lo, min(1.2, lo + 0.5)))
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
        out, ids, cnt = dec.call_batch(dev(cif[None]), 8, dev(caf[None]), 8)
        n = int(cnt[0])
        assert n == len(want), 'case %d (%dx%d, %d people): %d poses, oracle %d' % (case, H, W, people, n, len(want))
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert ok, 'case %d (%dx%d, %d people): %s' % (case, H, W, people, msg)


def test_large_annotation_capacity(native, port, coco_skeleton0):
    """max_annotations beyond what the keypoint-NMS scratch of eight waves fits in LDS: fewer waves share
    the NMS pass; the result does not change."""
    cif, caf = fields(95, 12)
    want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
    for cap in (700, 2000):
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=cap)
        out, ids, cnt = dec.call_batch(dev(cif[None]), 8, dev(caf[None]), 8)
        n = int(cnt[0])
        ok, msg = compare_annotations(out[0, :n].cpu().numpy(), want)
        assert n == len(want) and ok, (cap, msg)
