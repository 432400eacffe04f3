"""Seeds with EQUAL scores.  The reference sorts its seeds with an unstable ``std::sort`` (cif_seeds.cpp:94,118): where
scores tie, their order -- and with it which seed is grown first -- is what libstdc++'s introsort leaves.  The HIP
path reproduces that order for the images that have ties (cifseeds.hip: cifseeds_tie_kernel); these tests compare it
with the restatement, which calls the same ``std::sort`` (oracle/cifcaf_oracle.cpp, tie rule 0), and with the reference
itself."""
import numpy as np
import pytest

from common import compare_annotations, to_bf16

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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _seeds(native, cif, stride=8):
    hr = native.CifHr()
    hr.accumulate(dev(cif), stride)
    seeds = native.CifSeeds(hr)
    seeds.fill(dev(cif), stride)
    f, v = seeds.get()
    return f.cpu().numpy(), v.cpu().numpy()


def _quantised_field(seed, F, H, W, levels, active):
    """A CIF field of 4x4-cell blobs: the 16 cells of a blob regress to the blob's centre (their Gaussians stack there,
    like the cells around a keypoint), a fraction `active` of the blobs is on, and confidences take `levels` distinct
    values -- so the seed scores (0.9 map value at the centre + 0.1 confidence, cif_seeds.cpp:56) tie inside every blob
    and, for few levels, across blobs."""
    rng = np.random.default_rng(seed)
    cif = np.zeros((F, 5, H, W), dtype=np.float32)
    conf = (rng.integers(1, levels + 1, size=(F, H, W)) / levels * 0.95).astype(np.float32)
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    on = rng.random((F, (H + 3) // 4, (W + 3) // 4)) < active
    conf[~on[:, jj // 4, ii // 4]] = 0.0
    cif[:, 1] = conf
    cif[:, 2] = np.minimum(4 * (ii // 4) + 1.5, W - 1).astype(np.float32)
    cif[:, 3] = np.minimum(4 * (jj // 4) + 1.5, H - 1).astype(np.float32)
    cif[:, 4] = 1.0
    return cif


def _assert_same_seeds(native, port, cif, what):
    f, v = _seeds(native, cif)
    hr = port.cifhr_accumulate(cif, 8)
    want_f, want_v = port.cifseeds(cif, 8, hr)                    # tie rule 0: libstdc++'s std::sort
    assert len(f) == len(want_f), what
    assert len(np.unique(want_v[:, 0])) < len(want_v), '%s: the field was meant to produce equal scores' % what
    assert np.array_equal(f, want_f), '%s: fields differ first at seed %d of %d' % (
        what, int(np.argmax(f != want_f)), len(f))
    assert np.array_equal(v, want_v), '%s: seed rows differ first at %d' % (
        what, int(np.argmax((v != want_v).any(axis=1))))
    return len(f)


def test_seed_order_of_equal_scores_is_libstdcxx(native, port):
    """Stage level, sizes on both sides of every branch: one LDS block, the split sort (> 2 048), the global arrays
    of the tie pass (> 8 192) and the single-workgroup network (> 65 536 seeds, every cell of every field)."""
    assert native.get_seed_tie_order() == 'libstdcxx'
    sizes = []
    for seed, (F, H, W, levels, active) in enumerate([
            (1, 4, 4, 2, 1.0),           # 16 seeds: insertion sort only
            (3, 9, 11, 2, 0.3),          # a few dozen: the first partitions
            (5, 21, 23, 3, 0.5),         # ~1 000
            (17, 41, 41, 4, 0.25),       # ~7 000: LDS arrays, split first sort
            (17, 81, 81, 7, 0.2),        # ~22 000: global arrays
            (17, 81, 81, 1, 1.0),        # 111 537 seeds, few distinct scores
            (17, 81, 81, 200, 1.0)]):    # 111 537 seeds, many small groups
        cif = _quantised_field(100 + seed, F, H, W, levels, active)
        sizes.append(_assert_same_seeds(native, port, cif, 'field %d' % seed))
    assert min(sizes) <= 16 and max(sizes) > 65536 and any(2048 < n <= 8192 for n in sizes) and \
        any(8192 < n <= 65536 for n in sizes), sizes


def test_bf16_fields_seed_order_and_tie_state(native, port):
    """The fields of the bench's bfloat16 leg: every image has ties, the workspace says which were re-sorted."""
    from openpifpaf_amd import constants, synth
    cifs, cafs = synth.synth_batch(8, seed0=50_000)
    cifs, cafs = to_bf16(cifs), to_bf16(cafs)
    for b in range(8):
        _assert_same_seeds(native, port, cifs[b], 'image %d' % b)
    skel0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
    dec = native.CifCaf(17, torch.from_numpy(skel0))
    dec.call_batch(dev(cifs), 8, dev(cafs), 8)
    state = dec.workspace_view('seed_ties', torch.int32)[:8].cpu().numpy()
    assert (state == 1).all(), state
    clean, clean_caf = synth.synth_batch(8, seed0=50_000)         # float32 fields: equal scores are the exception
    dec.call_batch(dev(clean), 8, dev(clean_caf), 8)
    state = dec.workspace_view('seed_ties', torch.int32)[:8].cpu().numpy()
    for b in range(8):                                            # the flag is exactly "the sorted scores have equal neighbours"
        v = port.cifseeds(clean[b], 8, port.cifhr_accumulate(clean[b], 8))[1][:, 0]
        assert state[b] == int(len(np.unique(v)) < len(v)), (b, state)
    assert (state == 0).sum() >= 6, state


def test_bf16_fields_decode_equals_the_reference(native, port, coco_skeleton0):
    """What round 2 measured as a 4 % mismatch rate: tie-heavy fields, decoded, against the reference itself."""
    from openpifpaf_amd import synth
    from oracle import reference
    B = 48
    cifs, cafs = synth.synth_batch(B, seed0=60_000)
    cifs, cafs = to_bf16(cifs), to_bf16(cafs)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    use_ref = reference.available()
    if use_ref:
        reference.reset_statics()
    for b in range(B):
        if use_ref:
            want = reference.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)[0]
        else:
            want = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)[0]
        ok, msg = compare_annotations(out[b, :native.count_rows(int(counts[b]))], want)
        assert ok, 'image %d: %s' % (b, msg)


def test_index_order_is_still_there(native, port, coco_skeleton0):
    """``set_seed_tie_order('index')``: the total order of rounds 1-2 (score, then cell index), one launch less."""
    from openpifpaf_amd import synth
    cifs, cafs = synth.synth_batch(4, seed0=50_000)
    cifs, cafs = to_bf16(cifs), to_bf16(cafs)
    try:
        native.set_seed_tie_order('index')
        port.set_seed_tie_rule(1)
        for b in range(4):
            f, v = _seeds(native, cifs[b])
            hr = port.cifhr_accumulate(cifs[b], 8)
            want_f, want_v = port.cifseeds(cifs[b], 8, hr)
            assert np.array_equal(f, want_f) and np.array_equal(v, want_v)
    finally:
        native.set_seed_tie_order('libstdcxx')
        port.set_seed_tie_rule(0)


def test_wholebody_bf16_fields(native, port):
    """133 fields, ~20 000 seeds per image: the tie pass on its global arrays, through the whole decode."""
    from openpifpaf_amd import constants, synth
    wb = constants.wholebody()
    skel0 = np.asarray(wb['skeleton'], dtype=np.int64) - 1
    cifs, cafs = synth.synth_batch(2, seed0=7, people=(3, 6), pose=wb['standing_pose'], skeleton=wb['skeleton'])
    cifs, cafs = to_bf16(cifs), to_bf16(cafs)
    for b in range(2):
        _assert_same_seeds(native, port, cifs[b], 'wholebody image %d' % b)
    dec = native.CifCaf(133, torch.from_numpy(skel0))
    out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    for b in range(2):
        want = port.decode(cifs[b], 8, cafs[b], 8, skel0)[0]
        ok, msg = compare_annotations(out[b, :native.count_rows(int(counts[b]))], want)
        assert ok, 'image %d: %s' % (b, msg)


def test_tie_pass_inside_the_association_kernel_equals_the_separate_launch(native, port, coco_skeleton0):
    """Seed tie order 'libstdcxx-fused' (round 4): the pass that puts equal scores into std::sort's order runs in the
    association kernel's prologue, each image its own, instead of as a launch of its own.  Same seeds, same annotations,
    bit for bit -- on float32 fields with a few ties, on bf16-rounded ones with ties everywhere, and on a wholebody image
    beyond the LDS arrays (arrays in global memory, 12 waves instead of 16)."""
    from openpifpaf_amd import constants, synth
    cifs, cafs = synth.synth_batch(8, seed0=0)
    cases = [(cifs, cafs, coco_skeleton0), (to_bf16(cifs), to_bf16(cafs), coco_skeleton0)]
    wb = constants.wholebody()
    wc, wf = synth.synth_batch(2, seed0=3, people=(6, 10), pose=wb['standing_pose'], skeleton=wb['skeleton'])
    cases.append((to_bf16(wc), to_bf16(wf), np.asarray(wb['skeleton'], dtype=np.int64) - 1))
    old = native.get_seed_tie_order()
    try:
        for cif, caf, skel0 in cases:
            res = {}
            for mode in ('libstdcxx', 'libstdcxx-fused'):
                native.set_seed_tie_order(mode)
                assert native.get_seed_tie_order() == mode
                dec = native.CifCaf(cif.shape[1], torch.from_numpy(skel0))
                dec.set_tie_placement(mode == 'libstdcxx-fused')     # (round 6: a decoder's own choice; automatic = inside)
                out, ids, counts = dec.call_batch(dev(cif), 8, dev(caf), 8)
                counts = counts.cpu().numpy()
                native.check_counts(counts)
                B = len(counts)
                n_seeds = dec.workspace_view('seed_count', torch.int32)[:B].cpu().numpy()
                cap = cif.shape[1] * cif.shape[3] * cif.shape[4]
                sf = dec.workspace_view('seed_f', torch.int32)[:B * cap].view(B, cap).cpu().numpy()
                sv = dec.workspace_view('seed_vxys', torch.float32)[:B * cap * 4].view(B, cap, 4).cpu().numpy()
                ties = dec.workspace_view('seed_ties', torch.int32)[:B].cpu().numpy()
                res[mode] = (out.cpu().numpy(), counts, [sf[b, :n_seeds[b]].copy() for b in range(B)],
                             [sv[b, :n_seeds[b]].copy() for b in range(B)], ties)
            a, b_ = res['libstdcxx'], res['libstdcxx-fused']
            assert np.array_equal(a[1], b_[1]) and np.array_equal(a[4], b_[4]) and (a[4] != -1).all()
            for i in range(len(a[1])):
                assert np.array_equal(a[2][i], b_[2][i]) and np.array_equal(a[3][i], b_[3][i]), 'seed order of image %d' % i
                n = native.count_rows(int(a[1][i]))
                assert np.array_equal(a[0][i, :n], b_[0][i, :n])
            assert (a[4] == 1).any()                                   # the case has images with equal scores
        # and against the oracle (same std::sort) for the float32 batch
        native.set_seed_tie_order('libstdcxx-fused')
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
        out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
        counts = counts.cpu().numpy()
        for i in range(8):
            want, _ = port.decode(cifs[i], 8, cafs[i], 8, coco_skeleton0)
            ok, msg = compare_annotations(out[i, :native.count_rows(int(counts[i]))].cpu().numpy(), want)
            assert ok, msg
    finally:
        native.set_seed_tie_order(old)


def test_tie_placement_per_decoder_is_bit_identical(native, port, coco_skeleton0):
    """``opa_cifcaf_set_tie_placement`` (round 5): the pass that orders equal scores runs as a launch of its own or inside
    that decoder's association kernel -- a property of the decoder handle, whatever the process-wide tie order says; two or
    more ``DecodeLanes`` choose "inside".  Same seeds, same poses, bit for bit, on fields full of equal scores."""
    from openpifpaf_amd import synth
    cifs, cafs = synth.synth_batch(6, seed0=77_000)
    cifs, cafs = to_bf16(cifs), to_bf16(cafs)
    res = {}
    for inside in (False, True, None):
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
        dec.set_tie_placement(inside)
        out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
        state = dec.workspace_view('seed_ties', torch.int32)[:6].cpu().numpy()
        assert (state == 1).all(), (inside, state)
        res[inside] = (out.cpu().numpy(), counts.cpu().numpy(),
                       dec.workspace_view('seed_vxys', torch.float32)[:64].cpu().numpy())
    for k in (True, None):
        assert np.array_equal(res[False][1], res[k][1]) and np.array_equal(res[False][2], res[k][2])
        for b in range(6):
            n = native.count_rows(int(res[False][1][b]))
            assert np.array_equal(res[False][0][b, :n], res[k][0][b, :n])
    lanes = native.DecodeLanes(17, torch.from_numpy(coco_skeleton0), lanes=2)
    t = [lanes.submit(dev(cifs), 8, dev(cafs), 8) for _ in range(2)]
    for ticket in t:
        out, ids, counts = ticket.result()
        assert np.array_equal(counts.cpu().numpy(), res[False][1])
    for b in range(6):
        want = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)[0]
        ok, msg = compare_annotations(res[True][0][b, :native.count_rows(int(res[True][1][b]))], want)
        assert ok, (b, msg)
