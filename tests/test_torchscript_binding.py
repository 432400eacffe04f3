"""TorchScript custom-class binding (csrc/torch_binding.cpp): same class/method/static names as the
reference's registrations (csrc/src/module.cpp:19-118) under the openpifpaf_amd* namespaces."""
import io

import numpy as np
import pytest

torch = pytest.importorskip('torch')

# (class, [(name, value to set, type)]) -- one entry per STATIC_GETSET of module.cpp
STATICS = {
    ('openpifpaf_amd_decoder', 'CifCaf'): [
        ('block_joints', True), ('greedy', True), ('keypoint_threshold', 0.3), ('keypoint_threshold_rel', 0.4),
        ('reverse_match', False), ('force_complete', True), ('force_complete_caf_th', 0.01)],
    ('openpifpaf_amd_decoder_utils', 'CifHr'): [('neighbors', 12), ('threshold', 0.25), ('ablation_skip', True)],
    ('openpifpaf_amd_decoder_utils', 'CifSeeds'): [
        ('threshold', 0.35), ('ablation_nms', True), ('ablation_no_rescore', True)],
    ('openpifpaf_amd_decoder_utils', 'CifDetSeeds'): [('threshold', 0.35)],
    ('openpifpaf_amd_decoder_utils', 'CafScored'): [('default_score_th', 0.2), ('ablation_no_rescore', True)],
    ('openpifpaf_amd_decoder_utils', 'NMSKeypoints'): [
        ('instance_threshold', 0.2), ('keypoint_threshold', 0.25), ('suppression', 1e-4)],
}


def test_registers_reference_surface_and_shares_the_process_globals():
    from openpifpaf_amd import _lib, native, torchscript
    torchscript.load()
    before = _lib.get_params()
    try:
        for (ns, cls), fields in STATICS.items():
            c = getattr(getattr(torch.classes, ns), cls)
            for name, value in fields:
                old = getattr(c, 'get_' + name)()
                getattr(c, 'set_' + name)(value)
                assert getattr(c, 'get_' + name)() == value, (cls, name)
                getattr(c, 'set_' + name)(old)
        # the binding and the ctypes host mirror talk to ONE set of process globals
        torch.classes.openpifpaf_amd_decoder.CifCaf.set_keypoint_threshold(0.45)
        assert native.CifCaf.get_keypoint_threshold() == 0.45
        native.CifSeeds.set_threshold(0.33)
        assert torch.classes.openpifpaf_amd_decoder_utils.CifSeeds.get_threshold() == 0.33
    finally:
        _lib.set_params(before)
    torch.ops.openpifpaf_amd.set_quiet(True)
    torch.ops.openpifpaf_amd.set_quiet(False)


def test_constructor_fails_loudly_without_gpu_and_checks_dtype():
    from openpifpaf_amd import torchscript
    C = torchscript.load().CifCaf
    with pytest.raises(RuntimeError, match='LongTensor'):
        C(17, torch.zeros((19, 2), dtype=torch.int32))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no HIP device'):
            C(17, torch.zeros((19, 2), dtype=torch.int64))


def test_occupancy_class_has_the_reference_semantics():
    """openpifpaf_amd_decoder_utils.Occupancy (module.cpp:67-73, occupancy.cpp:13-79): a host-side map; checked
    against the package's independent Python restatement on random boxes (and against the real class in
    test_oracle_vs_reference.py)."""
    from openpifpaf_amd import torchscript, tracking
    torchscript.load()
    rng = np.random.default_rng(5)
    for reduction, min_scale in ((2.0, 4.0), (1.0, 0.1), (3.0, 2.0)):
        occ = torch.classes.openpifpaf_amd_decoder_utils.Occupancy(reduction, min_scale)
        want = tracking.Occupancy(reduction, min_scale)
        assert occ.get(1, 3.0, 3.0) and not occ.get(0, 3.0, 3.0)          # before reset: one empty [1,1,1] map
        for shape in ((3, 41, 57), (5, 20, 20), (3, 41, 57)):
            occ.reset(list(shape)); want.reset(shape)
            for _ in range(3):
                for _ in range(25):
                    f = int(rng.integers(shape[0]))
                    x, y = rng.uniform(-10, shape[2] + 10), rng.uniform(-10, shape[1] + 10)
                    sigma = rng.uniform(0.0, 9.0)
                    occ.set(f, x, y, sigma); want.set(f, x, y, sigma)
                for _ in range(300):
                    f = int(rng.integers(shape[0] + 2))
                    x, y = rng.uniform(-10, shape[2] + 10), rng.uniform(-10, shape[1] + 10)
                    assert occ.get(f, x, y) == want.get(f, x, y), (reduction, shape, f, x, y)
                occ.clear(); want.clear()
                assert not occ.get(0, 5.0, 5.0)
    with pytest.raises(RuntimeError, match='out of range'):
        occ.set(99, 1.0, 1.0, 1.0)


@pytest.mark.gpu
def test_scripted_decoder_module_equals_native_and_oracle(coco_skeleton0, tmp_path):
    from openpifpaf_amd import headmeta, native, synth, torchscript
    from oracle import port
    cif_meta, caf_meta = headmeta.cocokp_metas()
    module = torch.jit.script(torchscript.DecoderModule(cif_meta, caf_meta))
    path = str(tmp_path / 'decoder.pt')
    module.save(path)                                   # exercises def_pickle
    module = torch.jit.load(path)
    ref_dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    for seed, people, size in ((31, 3, 41), (32, 6, 65)):
        cif, caf = synth.synth_fields(seed, people, height=size, width=size)
        cif_t, caf_t = torch.from_numpy(cif).cuda(), torch.from_numpy(caf).cuda()
        ann, ids = module(cif_t, caf_t)
        want_ann, want_ids = ref_dec.call(cif_t, 8, caf_t, 8)
        assert ann.is_cuda and torch.equal(ann, want_ann) and torch.equal(ids, want_ids)
        oracle_ann, _ = port.decode(cif, 8, caf, 8, coco_skeleton0)
        assert ann.shape == oracle_ann.shape and len(oracle_ann) >= 1
        assert np.abs(ann.cpu().numpy().astype(np.float64) - oracle_ann).max() <= 1e-4
        # CPU tensors in -> CPU tensors out, like the reference
        ann_c, ids_c = module(torch.from_numpy(cif), torch.from_numpy(caf))
        assert not ann_c.is_cuda and torch.equal(ann_c, ann.cpu())


@pytest.mark.gpu
def test_binding_batch_cifhr_initial_annotations_and_blend(coco_skeleton0):
    from openpifpaf_amd import native, synth, torchscript
    C = torchscript.load().CifCaf
    dec = C(17, torch.from_numpy(coco_skeleton0))
    ref_dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    hr0, rev0 = dec.get_cifhr()
    assert rev0 == 0.0 and hr0.numel() == 1
    cifs, cafs = synth.synth_batch(3, seed0=40, height=41, width=41, people=(2, 4))
    cif_t, caf_t = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
    out, ids, counts = dec.call_batch(cif_t, 8, caf_t, 8)
    want = ref_dec.call_batch(cif_t, 8, caf_t, 8)
    assert torch.equal(counts.cpu(), want[2].cpu())
    for b in range(3):
        n = int(counts[b])
        assert n >= 1 and torch.equal(out[b, :n], want[0][b, :n]) and torch.equal(ids[b, :n], want[1][b, :n])
    # get_cifhr: image 0 of the last call, revision 1.0 semantics
    dec.call(cif_t[0], 8, caf_t[0], 8)
    hr, rev = dec.get_cifhr()
    ref_dec.call(cif_t[0], 8, caf_t[0], 8)
    want_hr, want_rev = ref_dec.get_cifhr()
    assert rev == want_rev and torch.equal(hr, want_hr)
    # tracking entry point: previous poses are continued first and keep their ids
    ann, _ = dec.call(cif_t[1], 8, caf_t[1], 8)
    init = ann[:1].clone()
    init_ids = torch.tensor([77], dtype=torch.int64)
    got = dec.call_with_initial_annotations(cif_t[1], 8, caf_t[1], 8, init, init_ids)
    want = ref_dec.call_with_initial_annotations(cif_t[1], 8, caf_t[1], 8, init, init_ids)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and 77 in got[1].tolist()
    with pytest.raises(RuntimeError, match='initial_ids'):
        dec.call_with_initial_annotations(cif_t[1], 8, caf_t[1], 8, init, None)
    # free function
    rows = torch.tensor([[0.9, 10.0, 10.0, 20.0, 20.0, 2.0, 2.0], [0.5, 10.5, 10.0, 22.0, 20.0, 2.0, 2.0]])
    got = torch.ops.openpifpaf_amd_decoder.grow_connection_blend(rows, 10.0, 10.0, 4.0, 1.0, False)
    want = native.grow_connection_blend(rows, 10.0, 10.0, 4.0, 1.0, False)
    assert list(got) == list(want)


@pytest.mark.gpu
def test_binding_pool_setting_survives_pickle_and_call_batch_recovers(coco_skeleton0, tmp_path):
    """scripted and batched users had no way to choose the CIF map's tile pool, and ``call_batch`` could
    not recover from an overflow.  The class now has ``set_cifhr_pool_tiles`` / ``get_cifhr_pool_tiles`` / ``use_full_pool``,
    the setting (and ``max_annotations``) is part of the pickle state, and ``call_batch`` decodes again with a full pool when
    an image comes back flagged -- here: TWO structureless all-active images in one batch, more than pool + spill region hold."""
    from openpifpaf_amd import native, synth, torchscript
    from oracle import port
    C = torchscript.load().CifCaf
    dec = C(17, torch.from_numpy(coco_skeleton0))
    assert dec.get_cifhr_pool_tiles() == 0
    dec.set_cifhr_pool_tiles(2048)
    dec.set_max_annotations(96)

    class Holder(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.d = d

        def forward(self, cif, caf):
            return self.d.call_batch(cif, 8, caf, 8)
    path = str(tmp_path / 'holder.pt')
    torch.jit.script(Holder(dec)).save(path)
    loaded = torch.jit.load(path)
    assert loaded.d.get_cifhr_pool_tiles() == 2048
    cifs, cafs = synth.synth_batch(2, seed0=70, height=41, width=41, people=(2, 3))
    out, ids, counts = loaded(torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda())
    assert out.shape[1] == 96 and not native.count_failed(counts.cpu().numpy()).any()
    # the automatic pool: two all-active images overflow pool + spill region, call_batch notices and repeats the decode
    auto = C(17, torch.from_numpy(coco_skeleton0))
    pairs = [synth.adversarial_fields(21 + i) for i in range(2)]
    cif_t = torch.from_numpy(np.stack([c for c, _ in pairs])).cuda()
    caf_t = torch.from_numpy(np.stack([f for _, f in pairs])).cuda()
    out, ids, counts = auto.call_batch(cif_t, 8, caf_t, 8)
    counts = counts.cpu().numpy()
    assert not native.count_failed(counts).any() and auto.get_cifhr_pool_tiles() == -1
    for b in range(2):
        want, _ = port.decode(pairs[b][0], 8, pairs[b][1], 8, coco_skeleton0)
        assert native.count_rows(int(counts[b])) == len(want)
    auto.use_full_pool()
    assert auto.get_cifhr_pool_tiles() == -1


@pytest.mark.gpu
def test_binding_stage_objects_and_cifdet_equal_the_ctypes_mirror(coco_skeleton0):
    """openpifpaf_amd_decoder_utils.{CifHr,CifSeeds,CafScored} and openpifpaf_amd_decoder.CifDet used the way
    the reference's tests/tools use theirs (module.cpp:57-111)."""
    from openpifpaf_amd import native, synth, torchscript
    from oracle import port
    torchscript.load()
    U = torch.classes.openpifpaf_amd_decoder_utils
    cif, caf = synth.synth_fields(50, 4, height=41, width=41)
    cif_t, caf_t = torch.from_numpy(cif).cuda(), torch.from_numpy(caf).cuda()
    skel = torch.from_numpy(coco_skeleton0)
    # reference call sequence: reset, accumulate, get_accumulated -> seeds / caf lists built from the map
    hr = U.CifHr()
    hr.reset(list(cif.shape), 8)
    hr.accumulate(cif_t, 8, 0.0, 1.0)
    acc, rev = hr.get_accumulated()
    want_hr = native.CifHr()
    want_hr.accumulate(cif_t, 8)
    assert rev == 1.0 and torch.equal(acc, want_hr.get_accumulated()[0])
    seeds = U.CifSeeds(acc, rev)
    seeds.fill(cif_t, 8)
    f, vxys = seeds.get()
    want_seeds = native.CifSeeds(want_hr)
    want_seeds.fill(cif_t, 8)
    wf, wv = want_seeds.get()
    assert len(f) > 0 and torch.equal(f, wf) and torch.equal(vxys, wv)
    # a plain contiguous copy of the map (not the pitched view) must work too
    seeds2 = U.CifSeeds(acc.contiguous().cpu(), rev)
    seeds2.fill(cif_t, 8)
    assert torch.equal(seeds2.get()[1], wv)
    scored = U.CafScored(acc, rev, -1.0, 0.1)
    scored.fill(caf_t, 8, skel)
    fwd, bwd = scored.get()
    want_scored = native.CafScored(want_hr, cif.shape, 8)
    want_scored.fill(caf_t, 8, skel)
    wfwd, wbwd = want_scored.get()
    assert len(fwd) == 19 and sum(len(x) for x in fwd) > 0
    assert all(torch.equal(a, b) for a, b in zip(fwd, wfwd)) and all(torch.equal(a, b) for a, b in zip(bwd, wbwd))
    with pytest.raises(RuntimeError, match='revision'):
        U.CifSeeds(acc, 2.0)
    # CifDetSeeds on the detection map of the oracle (the reference exports no CifDetHr object either)
    field_np = synth.synth_det_field(3, 5, height=33, width=41)
    _, _, _, det_hr = port.cifdet_decode(field_np, 8, return_cifhr=True)
    want_f, want_v = port.cifdetseeds(field_np, 8, det_hr)
    for seeds_obj in (U.CifDetSeeds(torch.from_numpy(det_hr), 1.0), native.CifDetSeeds(torch.from_numpy(det_hr))):
        seeds_obj.fill(torch.from_numpy(field_np).cuda(), 8)
        got_f, got_v = seeds_obj.get()
        assert len(want_f) > 0 and got_v.shape == (len(want_f), 5)
        assert np.array_equal(got_f.cpu().numpy(), want_f) and np.array_equal(got_v.cpu().numpy(), want_v)
    # CifDet
    D = torch.classes.openpifpaf_amd_decoder.CifDet
    assert D.get_max_detections_before_nms() == 120
    field = torch.from_numpy(synth.synth_det_field(3, 5, height=33, width=41))
    cat, sc, bx = D().call(field.cuda(), 8)
    wcat, wsc, wbx = native.CifDet().call(field.cuda(), 8)
    assert len(cat) > 0 and torch.equal(cat, wcat) and torch.equal(sc, wsc) and torch.equal(bx, wbx)
    cat_c, _, _ = D().call(field, 8)
    assert not cat_c.is_cuda and torch.equal(cat_c, cat.cpu())
