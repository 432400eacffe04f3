"""Round-6 parity cases: every structural change of the round decodes exactly like the structure it replaced and like the oracle --
the tile work list / seed candidates against round 5's stage kernels, the image queue of the association kernel (batches of more
images than compute units), the side stream, the one-pass CAF list sets, and ``opa_debug`` itself (per decoder, no environment)."""
import numpy as np
import pytest

from common import compare_annotations

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

FC_KW = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
             nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)       # reference decoder/cifcaf.py:180-185


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


def decode(native, skeleton0, cifs, cafs, params=None, debug=None, **kw):
    dec = native.CifCaf(cifs.shape[1], torch.from_numpy(skeleton0), **kw)
    if debug:
        dec.set_debug(**debug)
    out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8, params=params)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    native.check_counts(counts)
    return [out[b, :native.count_rows(int(counts[b]))] for b in range(len(counts))], dec


def test_debug_struct_is_per_decoder_and_nothing_reads_the_environment(native, coco_skeleton0, monkeypatch):
    """``opa_debug`` (round 6): the switches of rounds 2-5 were environment variables looked up at every launch.  They are a struct
    of the decoder handle now; the environment is read once, when the library is loaded -- changing it afterwards changes nothing."""
    from openpifpaf_amd import _lib, synth
    d0 = _lib.default_debug()
    assert d0.stage_worklist == 1 and d0.assoc_growers == 0 and d0.assoc_watchdog_ticks == 100000000
    a = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    b = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    a.set_debug(assoc_growers=3, assoc_predict=0)
    assert a.get_debug().assoc_growers == 3 and a.get_debug().assoc_predict == 0
    assert b.get_debug().assoc_growers == 0 and b.get_debug().assoc_predict == 1, 'a decoder\'s switches are its own'
    with pytest.raises(AttributeError):
        a.set_debug(no_such_switch=1)
    with pytest.raises(_lib.NativeError):
        a.set_debug(assoc_watchdog_ticks=0)
    cifs, cafs = synth.synth_batch(2, seed0=91_000, height=41, width=41)
    monkeypatch.setenv('OPA_ASSOC_GROWERS', '1')          # (rounds 2-5: this changed the very next launch)
    monkeypatch.setenv('OPA_ASSOC_WATCHDOG_TICKS', '1')
    out, ids, counts = b.call_batch(dev(cifs), 8, dev(cafs), 8)
    native.check_counts(counts)                           # no watchdog failure
    assert (b.assoc_stats()[:, 13].cpu().numpy() == 11).all(), 'the launch took its growers from the environment'
    out, ids, counts = a.call_batch(dev(cifs), 8, dev(cafs), 8)
    assert (a.assoc_stats()[:, 13].cpu().numpy() == 3).all()
    a.set_debug()                                         # back to the defaults
    assert a.get_debug().assoc_growers == 0


def test_work_list_tiles_and_seed_candidates_equal_round_5s_stage_kernels(native, port, coco_skeleton0):
    """The map through the tile work list (one tile per wave, three-instruction exact quotient) and the seeds from the
    candidate lists against round 5's kernels (``stage_worklist = 0``): the gathered map, the seeds in std::sort's order, the
    annotations -- bit for bit; and both against the oracle.  Crowded, noisy and empty images, a non-square field."""
    from openpifpaf_amd import synth
    cases = [synth.synth_fields(60_000 + i, p, height=57, width=73, cif_noise=n) for i, (p, n) in
             enumerate([(0, None), (1, None), (7, None), (15, 0.5), (24, None), (9, 0.9)])]
    cifs, cafs = np.stack([c for c, _ in cases]), np.stack([f for _, f in cases])
    new, dn = decode(native, coco_skeleton0, cifs, cafs)
    old, do = decode(native, coco_skeleton0, cifs, cafs, debug={'stage_worklist': 0})
    B, cap = len(cases), 17 * 57 * 73
    for b in range(B):
        assert np.array_equal(dn.get_cifhr(b)[0].cpu().numpy(), do.get_cifhr(b)[0].cpu().numpy()), 'map of image %d' % b
        want_hr = port.cifhr_accumulate(cifs[b], 8)
        assert np.array_equal(dn.get_cifhr(b)[0].cpu().numpy() != 0, want_hr != 0)
    n_new = dn.workspace_view('seed_count', torch.int32)[:B].cpu().numpy()
    n_old = do.workspace_view('seed_count', torch.int32)[:B].cpu().numpy()
    assert np.array_equal(n_new, n_old) and n_new[0] == 0 and n_new.max() > 1000
    sv_new = dn.workspace_view('seed_vxys', torch.float32)[:B * cap * 4].view(B, cap, 4).cpu().numpy()
    sv_old = do.workspace_view('seed_vxys', torch.float32)[:B * cap * 4].view(B, cap, 4).cpu().numpy()
    for b in range(B):
        assert np.array_equal(sv_new[b, :n_new[b]], sv_old[b, :n_old[b]]), 'seeds of image %d' % b
        assert np.array_equal(new[b], old[b]), 'annotations of image %d' % b
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        ok, msg = compare_annotations(new[b], want)
        assert ok, 'image %d: %s' % (b, msg)
    # the seed-rescoring ablations go through the candidate lists too (cif_seeds.cpp:35-40,49-53)
    from openpifpaf_amd import _lib
    for kw in (dict(ablation_cifseeds_nms=1), dict(ablation_cifseeds_no_rescore=1), dict(seed_threshold=0.35, cif_threshold=0.25)):
        got, _ = decode(native, coco_skeleton0, cifs, cafs, params=_lib.default_params(**kw))
        for b in range(B):
            want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0, params=port.default_params(**kw))
            ok, msg = compare_annotations(got[b], want)
            assert ok, '%r image %d: %s' % (kw, b, msg)


def test_image_queue_of_large_batches_is_longest_first_and_exact(native, port, coco_skeleton0):
    """More images than the chip has compute units (round 6): the association workgroups take their images from a queue ordered by
    seed count, most seeds first.  300 small images: the queue's order, every image decoded exactly once, bit-identical with
    workgroup b = image b, a sample against the oracle; and the queue forced on for a small batch."""
    from openpifpaf_amd import synth
    B = 300
    rng = np.random.default_rng(5)
    people = rng.integers(0, 9, B)
    cases = [synth.synth_fields(61_000 + i, int(people[i]), height=33, width=41) for i in range(B)]
    cifs, cafs = np.stack([c for c, _ in cases]), np.stack([f for _, f in cases])
    got, dec = decode(native, coco_skeleton0, cifs, cafs)
    assert torch.cuda.get_device_properties(0).multi_processor_count < B
    order = dec.workspace_view('assoc_queue', torch.int32)[:B + 1].cpu().numpy()
    seeds = dec.workspace_view('seed_count', torch.int32)[:B].cpu().numpy()
    assert sorted(order[:B].tolist()) == list(range(B)) and order[B] == B, 'every image once, the head at the end'
    assert (np.diff(seeds[order[:B]]) <= 0).all(), 'most seeds first'
    plain, _ = decode(native, coco_skeleton0, cifs, cafs, debug={'assoc_persistent': -1})
    for b in range(B):
        assert np.array_equal(got[b], plain[b]), 'image %d changes with the queue' % b
    for b in range(0, B, 13):
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        ok, msg = compare_annotations(got[b], want)
        assert ok, 'image %d: %s' % (b, msg)
    small, _ = decode(native, coco_skeleton0, cifs[:9], cafs[:9], debug={'assoc_persistent': 1})
    for b in range(9):
        assert np.array_equal(small[b], got[b])


def test_side_stream_and_list_set_passes_are_bit_identical(native, port, coco_skeleton0):
    """The CAF lists built on the handle's side stream beside the seed chain (``side_stream``), and both list sets of a
    force-complete decode from one read of the field or from two passes (``scored_one_pass``): the same annotations, several calls
    in a row on one decoder (the fork / join events are re-used), default flags and the reference benchmark's setting."""
    from openpifpaf_amd import _lib, synth
    cifs, cafs = synth.synth_batch(6, seed0=62_000, height=49, width=65)
    for kw in ({}, FC_KW):
        params = _lib.default_params(**kw) if kw else None
        want, _ = decode(native, coco_skeleton0, cifs, cafs, params=params)
        for b in range(6):
            ref, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0, params=port.default_params(**kw) if kw else None)
            ok, msg = compare_annotations(want[b], ref)
            assert ok, msg
        for debug in ({'side_stream': 1}, {'scored_one_pass': 0}, {'side_stream': 1, 'scored_one_pass': 0}):
            dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
            dec.set_debug(**debug)
            for rep in range(3):
                out, ids, counts = dec.call_batch(dev(cifs), 8, dev(cafs), 8, params=params)
                counts = counts.cpu().numpy()
                native.check_counts(counts)
                for b in range(6):
                    assert np.array_equal(out[b, :native.count_rows(int(counts[b]))].cpu().numpy(), want[b]), (kw, debug, rep, b)


def test_rccl_gather_branch_runs_on_one_gpu(coco_skeleton0, tmp_path):
    """The job's one collective, ``dist.all_gather_into_tensor`` on DEVICE tensors through backend "nccl" (= RCCL), had never
    run: no multi-GPU box, and a group of one rank returns early.  Here a process group of ONE rank is initialised with nccl and
    the collective is forced (``gather_annotations(force=True)``): packed annotation blocks of a real decode go through RCCL and
    come back bit for bit.  (In a process of its own: a process group is process-wide state.)"""
    import os
    import socket
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'rccl_one_rank.py'
    script.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        from openpifpaf_amd import constants, distributed as D, native, synth
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
        assert dist.get_world_size() == 1 and dist.get_backend() == 'nccl'
        skel0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
        cifs, cafs = synth.synth_batch(5, seed0=63_000, height=41, width=49)
        dec = native.CifCaf(17, torch.from_numpy(skel0))
        out, ids, counts = dec.call_batch(torch.from_numpy(cifs).cuda(), 8, torch.from_numpy(cafs).cuda(), 8)
        native.check_counts(counts)
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
        a, i, c = D.gather_annotations(out, ids, counts, force=True)
        dist.all_gather_into_tensor = real
        torch.cuda.synchronize()
        assert len(calls) == 1 and a.is_cuda
        assert torch.equal(c, counts) and torch.equal(i, ids)
        n = int(native.count_rows(int(counts.max())))
        assert n > 0 and torch.equal(a[:, :n].view(torch.int32), out[:, :n].view(torch.int32))
        a0, i0, c0 = D.gather_annotations(out, ids, counts)          # without force: a group of one rank is the identity
        assert a0 is out and len(calls) == 1
        per_image = D.unpack(a, i, c)
        assert sum(len(p) for p, _ in per_image) == int(sum(native.count_rows(int(x)) for x in counts.cpu()))
        dist.destroy_process_group()
        print('RCCL_OK')
    ''' % root))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
