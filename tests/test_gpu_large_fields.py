"""Fields larger than the BASELINE's 641 px (missing 3): the reference takes any ``--long-edge``
(predictor.py:85-102); at 1281 px the fields are 161 x 161.  What changes with the size and is not exercised by the
COCO shapes: a key array of more than 2^17 entries, more than 256 tiles per CifHr plane (861 here), more than 8192 seeds
per image (the rank merge over several sort blocks and, with equal scores, the tie pass on arrays in global memory),
lists of more than 100 chunks, occupancy maps wider than 512 cells.  Every stage bit-exact against the oracle, the decode
within the north-star tolerance, default flags and the reference benchmark's force-complete setting."""
import numpy as np
import pytest

from common import compare_annotations, to_bf16

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

FC_KW = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
             nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)       # reference decoder/cifcaf.py:180-185

# (seed, people, H, W, person size range): 1281 x 1281 and 961 x 1281 (H x W) images
LARGE = [(9001, 40, 161, 161, (0.2, 0.6)), (9002, 25, 121, 161, (0.3, 0.8)), (9004, 45, 161, 161, (0.35, 0.8)),
         (9005, 90, 161, 161, (0.2, 0.5))]


@pytest.fixture(scope='module')
def native():
    from openpifpaf_amd import native as n
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return n


@pytest.fixture(scope='module')
def port():
    from oracle import port as p
    return p


_cache = {}


def fields(case):
    from openpifpaf_amd import synth
    if case not in _cache:
        seed, people, H, W, sr = case
        _cache[case] = synth.synth_fields(seed, people, height=H, width=W, size_range=sr)
    return _cache[case]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('case', LARGE[:3])
def test_large_field_stages_are_bit_exact(native, port, coco_skeleton0, case):
    cif, caf = fields(case)
    H, W = cif.shape[-2:]
    assert ((H - 1) * 8 + 1 + 31) // 32 * (((W - 1) * 8 + 1 + 63) // 64) > 256        # tiles per plane
    ref_hr = port.cifhr_accumulate(cif, 8)
    hr = native.CifHr()
    hr.accumulate(dev(cif), 8)
    got = hr.get_accumulated()[0].cpu().numpy()
    assert got.shape == ref_hr.shape and np.array_equal(got, ref_hr), '%d map cells differ' % (got != ref_hr).sum()
    ref_f, ref_v = port.cifseeds(cif, 8, ref_hr)
    seeds = native.CifSeeds(hr)
    seeds.fill(dev(cif), 8)
    f, v = seeds.get()
    f, v = f.cpu().numpy(), v.cpu().numpy()
    assert len(f) == len(ref_f) > 2048                                             # more than one sort block
    assert np.array_equal(f, ref_f) and np.array_equal(v, ref_v)                   # std::sort's order, ties included
    ref_fw, ref_bw = port.cafscored(caf, 8, ref_hr, cif.shape, 8, coco_skeleton0)
    cs = native.CafScored(hr, cif.shape, 8)
    cs.fill(dev(caf), 8, torch.from_numpy(coco_skeleton0))
    fwd, bwd = cs.get()
    for a in range(len(ref_fw)):
        assert np.array_equal(fwd[a].cpu().numpy(), ref_fw[a]) and np.array_equal(bwd[a].cpu().numpy(), ref_bw[a]), a


@pytest.mark.parametrize('fc', [False, True])
def test_large_field_batch_decodes_like_the_oracle(native, port, coco_skeleton0, fc):
    """Batches of 161 x 161 and of 121 x 161 fields through the C ABI's batched decode; one decoder per shape, called
    twice with different images (the lazily cleared map of 861-tile planes carries over)."""
    params_dev = None
    params_ref = None
    if fc:
        from openpifpaf_amd import _lib
        params_dev, params_ref = _lib.default_params(**FC_KW), port.default_params(**FC_KW)
    poses = 0
    for shape_cases in ([LARGE[0], LARGE[2], LARGE[3]], [LARGE[1]]):
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
        for order in (shape_cases, shape_cases[::-1]):
            cifs = np.stack([fields(c)[0] for c in order])
            cafs = np.stack([fields(c)[1] for c in order])
            out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8, params=params_dev)
            out, counts = out.cpu().numpy(), counts.cpu().numpy()
            if LARGE[3] in order and fc is False and order is shape_cases:
                # the 90-person image reaches ~2200 map tiles, the automatic pool of a 161 x 161 field holds 1829: the rest sits
                # in the batch's spill region, and the image decodes in the same call (round 4: flagged and decoded again).
                # A pool of exactly that size WITHOUT a spill region flags the image -- and only that one --, status -2
                assert not dec.pool_overflowed()
                tight = native.CifCaf(17, torch.from_numpy(coco_skeleton0), cifhr_pool_tiles=1829)
                _, _, tc = tight.call_batch(dev(cifs), 8, dev(cafs), 8, params=params_dev)
                tc = tc.cpu().numpy()
                bad = order.index(LARGE[3])
                assert native.count_failed(tc).tolist() == [i == bad for i in range(len(order))]
                status = tight.workspace_view('status', torch.int32)[:len(order)].cpu().numpy()
                assert status[bad] == -2 and tight.pool_overflowed()
                with pytest.raises(Exception):
                    native.check_counts(tc)
                del tight
            native.check_counts(counts)
            assert not native.count_overflowed(counts).any()
            for b, c in enumerate(order):
                want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0, params=params_ref)
                ok, msg = compare_annotations(out[b, :native.count_rows(int(counts[b]))], want)
                assert ok, 'case %s fc=%s: %s' % (c[:4], fc, msg)
                poses += len(want)
        n_seeds = dec.workspace_view('seed_count', torch.int32)[:len(order)].cpu().numpy()
        assert n_seeds.max() > 2048
    assert poses > 300


def test_large_field_ties_in_global_memory(native, port, coco_skeleton0):
    """bf16-rounded 161 x 161 fields: more than 8192 seeds per image, nearly all with tied scores -- the tie pass works on
    arrays in global memory (an image beyond its LDS arrays) and must leave the seeds in std::sort's order."""
    for case in (LARGE[2], LARGE[3]):
        cif, caf = (to_bf16(a) for a in fields(case))
        ref_hr = port.cifhr_accumulate(cif, 8)
        ref_f, ref_v = port.cifseeds(cif, 8, ref_hr)
        assert len(ref_f) > 8192 and len(np.unique(ref_v[:, 0])) < len(ref_v) - 1000           # thousands of tied seeds
        hr = native.CifHr()
        hr.accumulate(dev(cif), 8)
        seeds = native.CifSeeds(hr)
        seeds.fill(dev(cif), 8)
        f, v = seeds.get()
        assert np.array_equal(f.cpu().numpy(), ref_f) and np.array_equal(v.cpu().numpy(), ref_v)
        dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
        got, _ = dec.call(dev(cif), 8, dev(caf), 8)
        assert int(dec.workspace_view('seed_ties', torch.int32)[0]) == 1             # re-sorted in libstdc++'s order
        want, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
        ok, msg = compare_annotations(got.cpu().numpy(), want)
        assert ok, msg


def test_map_tile_pool_overflow_is_flagged_and_retried(native, port, coco_skeleton0):
    """The decode keeps the high-resolution CIF map as a pool of 32x64 tiles (opa_shape::cifhr_pool_tiles; automatic: 1024 of
    the 3927 tiles of a 641-px COCO image, plus one spill region for the batch that holds the rest of a whole map).
    Structureless all-active fields reach every tile: ONE such image in a batch decodes through the spill region (the
    reference never fails, cif_hr.cpp:97-121); with two of them the region runs out: the asynchronous batched call flags
    what did not fit (OPA_COUNT_FAILED, status -2) instead of decoding it wrongly, the synchronous entry points and the
    decoder layer decode again with a pool that holds the whole map; an explicit small pool (no spill region) overflows on
    an ordinary crowded image, a full pool never does -- and every result equals the oracle's."""
    from openpifpaf_amd import decoder, headmeta, synth
    adv_cif, adv_caf = synth.adversarial_fields(11)
    adv2_cif, adv2_caf = synth.adversarial_fields(12)
    ok_cif, ok_caf = synth.synth_fields(31, 20, height=81, width=81)
    cifs, cafs = np.stack([ok_cif, adv_cif]), np.stack([ok_caf, adv_caf])
    want = [port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)[0] for b in range(2)]
    cifs2, cafs2 = np.stack([adv_cif, adv2_cif]), np.stack([adv_caf, adv2_caf])
    want2 = [want[1], port.decode(adv2_cif, 8, adv2_caf, 8, coco_skeleton0)[0]]

    # one structureless image beside an ordinary one: its own pool + the spill region hold its whole map
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8)
    counts_h = counts.cpu().numpy()
    native.check_counts(counts_h)
    assert not dec.pool_overflowed()
    for b in range(2):
        okk, msg = compare_annotations(out[b, :native.count_rows(int(counts_h[b]))].cpu().numpy(), want[b])
        assert okk, (b, msg)
    hr, rev = dec.get_cifhr(1)
    assert rev == 1.0 and np.array_equal(hr.cpu().numpy(), port.cifhr_accumulate(adv_cif, 8))    # the whole map, through pool + spill slots
    bytes_auto = dec._last[1].numel()
    # two of them: what the spill region cannot hold is flagged, never decoded wrongly
    out, ids, counts = dec.call_batch(dev(cifs2), 8, dev(cafs2), 8)
    counts_h = counts.cpu().numpy()
    failed = native.count_failed(counts_h)
    assert failed.any() and dec.pool_overflowed()
    status = dec.workspace_view('status', torch.int32)[:2].cpu().numpy()
    for b in range(2):
        if failed[b]:
            assert status[b] == -2 and native.count_rows(int(counts_h[b])) == 0
        else:
            okk, msg = compare_annotations(out[b, :native.count_rows(int(counts_h[b]))].cpu().numpy(), want2[b])
            assert okk, (b, msg)
    with pytest.raises(Exception):
        native.check_counts(counts_h)
    # synchronous single-image call: B = 1, the spill region is the image's own
    dec1 = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    got, _ = dec1.call(dev(adv_cif), 8, dev(adv_caf), 8)
    assert dec1.cifhr_pool_tiles == 0
    okk, msg = compare_annotations(got.cpu().numpy(), want[1])
    assert okk, msg
    # the full pool on request: the dense map's size, never overflows
    full = native.CifCaf(17, torch.from_numpy(coco_skeleton0), cifhr_pool_tiles='full')
    out, ids, counts = full.call_batch(dev(cifs2), 8, dev(cafs2), 8)
    counts_h = counts.cpu().numpy()
    native.check_counts(counts_h)
    for b in range(2):
        okk, msg = compare_annotations(out[b, :native.count_rows(int(counts_h[b]))].cpu().numpy(), want2[b])
        assert okk, (b, msg)
    assert full._last[1].numel() - bytes_auto == (3927 - 1024) * 8192
    # an explicit small pool (no spill region): the 20-person image alone reaches ~450 tiles
    small = native.CifCaf(17, torch.from_numpy(coco_skeleton0), cifhr_pool_tiles=256)
    out, ids, counts = small.call_batch(dev(ok_cif[None]), 8, dev(ok_caf[None]), 8)
    assert native.count_failed(counts.cpu().numpy()).all() and small.pool_overflowed()
    # the decoder layer: synchronous and pipelined batches repeat the decode with a full pool
    old_workers = decoder.CifCaf.decoder_workers
    try:
        decoder.CifCaf.decoder_workers = 2
        class Heads:
            def __call__(self, images):
                return (dev(cifs2), dev(cafs2))
        d = decoder.factory(list(headmeta.cocokp_metas()))
        sync = d.batch(Heads(), torch.zeros((2, 3, 641, 641)), device=torch.device('cuda'))
        assert [len(r) for r in sync] == [len(w) for w in want2]
        d2 = decoder.factory(list(headmeta.cocokp_metas()))
        pend = [d2.batch_async(Heads(), torch.zeros((2, 3, 641, 641)), device=torch.device('cuda')) for _ in range(3)]
        for p in pend:
            assert [len(r) for r in p.result()] == [len(w) for w in want2]
        assert d2.decoders[0].cpp_decoder.cifhr_pool_tiles == -1
    finally:
        decoder.CifCaf.decoder_workers = old_workers
