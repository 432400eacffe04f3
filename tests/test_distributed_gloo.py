"""N>1 path on CPU: world_size=2 gloo processes shard a batch by image and gather
fixed-size annotation blocks (the one collective of the path)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from openpifpaf_amd import distributed as D
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    n_images, max_ann, K = 6, 4, 17
    lo, hi = D.shard_bounds(n_images, rank, world)
    assert hi - lo == 3
    # stand-in for the device decode of the local shard: deterministic per global image index
    ann = torch.zeros((hi - lo, max_ann, K, 4))
    ids = torch.full((hi - lo, max_ann), -1, dtype=torch.int64)
    counts = torch.zeros((hi - lo,), dtype=torch.int32)
    for i, g in enumerate(range(lo, hi)):
        counts[i] = g %% (max_ann + 1)
        ann[i, :int(counts[i])] = float(g + 1)
        ids[i, :int(counts[i])] = g
    calls = []
    real_all_gather = dist.all_gather
    dist.all_gather = lambda *a_, **k_: (calls.append(1), real_all_gather(*a_, **k_))[1]
    a, i_, c = D.gather_annotations(ann, ids, counts)
    dist.all_gather = real_all_gather
    assert len(calls) == 1, 'the annotations of a batch travel in ONE collective (SURVEY 8e), saw %%d' %% len(calls)
    assert torch.equal(i_[lo:hi], ids) and torch.equal(a[lo:hi], ann)
    assert a.shape == (n_images, max_ann, K, 4) and c.tolist() == [g %% (max_ann + 1) for g in range(n_images)]
    per_image = D.unpack(a, i_, c)
    for g, (pa, pi) in enumerate(per_image):
        assert len(pa) == g %% (max_ann + 1)
        assert (pa == g + 1).all() and (pi == g).all()
    # an image whose decode FAILED on one rank (OPA_COUNT_FAILED travels with its count) must not come out of the gather as
    # "nobody in the picture": unpack raises on every rank that looks at the gathered batch
    from openpifpaf_amd import native, _lib
    bad_counts = counts.clone()
    if rank == 1:
        bad_counts[0] = native.COUNT_FAILED
    a2, i2, c2 = D.gather_annotations(ann, ids, bad_counts)
    assert native.count_failed(c2.numpy()).tolist() == [g == 3 for g in range(n_images)]
    try:
        D.unpack(a2, i2, c2)
        raise SystemExit('rank %%d: a failed image passed for an empty one' %% rank)
    except _lib.NativeError as e:
        assert '[3]' in str(e), str(e)
    # every rank adopts rank 0's table of 1x1-convolution kernel choices (the paths round differently)
    from openpifpaf_amd import fused
    key = ('torch.float32', 32 * 81 * 81, 64, 256, True, False)
    fused.set_choices({key: 'gemm' if rank == 0 else 'conv', ('only-on', rank): 'conv'}, replace=True)
    table = D.broadcast_conv_choices()
    assert table == {key: 'gemm', ('only-on', 0): 'conv'} and fused.choices() == table, table
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
''') % ROOT


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_bounds_cover_everything():
    from openpifpaf_amd import distributed as D
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d ok' % rank in out


def test_gather_is_identity_without_process_group():
    import torch
    from openpifpaf_amd import distributed as D
    a, i, c = torch.zeros((2, 3, 17, 4)), torch.zeros((2, 3), dtype=torch.int64), torch.zeros((2,), dtype=torch.int32)
    out = D.gather_annotations(a, i, c)
    assert out[0] is a and out[1] is i and out[2] is c


def test_pack_roundtrip_is_bit_exact():
    import torch
    from openpifpaf_amd import distributed as D
    g = torch.Generator().manual_seed(0)
    a = torch.randn((4, 6, 17, 4), generator=g)
    a[0, 0, 0, 0] = float('nan')
    i = torch.randint(-3, 1 << 40, (4, 6), generator=g)
    c = torch.tensor([0, 6, 3, 0x40000002], dtype=torch.int32)
    block = D.pack(a, i, c)
    assert block.dtype == torch.int32 and block.shape == (4, 1 + 6 * (17 * 4 + 2))
    a2, i2, c2 = D.unpack_block(block, 6, 17)
    assert torch.equal(a.view(torch.int32), a2.view(torch.int32)) and torch.equal(i, i2) and torch.equal(c, c2)
    assert len(D.unpack(a2, i2, c2)[3][0]) == 2          # overflow bit masked: two valid rows


# ---- the PRODUCT path under torch.distributed: Predictor.numpy_images -> decoder.CifCaf.batch shards every batch over the ranks,
# decodes its shard, ONE gather, every rank has the whole batch (reference predictor.py:33-37: nn.DataParallel).  No GPU here: the
# device decode is stood in for by the oracle (decode_heads is the one method overridden); everything around it is the product code.
PRODUCT_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from openpifpaf_amd import decoder, headmeta, native, predictor, synth, distributed as D
    from oracle import port
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W, N = 33, 41, 7
    fields = [synth.synth_fields(400 + i, 1 + i %% 4, height=H, width=W) for i in range(N)]
    skel0 = None

    class OracleCifCaf(decoder.CifCaf):          # the device decode stood in for by the oracle: call_batch's result, packed the same way
        calls = 0
        def __init__(self, cif_metas, caf_metas):
            decoder.Decoder.__init__(self)
            self.cif_metas, self.caf_metas = cif_metas, caf_metas
            self.score_weights = cif_metas[0].score_weights
            self.confidence_scales = caf_metas[0].decoder_confidence_scales
        def decode_heads(self, heads):
            OracleCifCaf.calls += 1
            cif, caf = heads[0].numpy(), heads[1].numpy()
            B, M, K = len(cif), 16, 17
            out, ids, counts = np.zeros((B, M, K, 4), np.float32), np.full((B, M), -1, np.int64), np.zeros((B,), np.int32)
            skel = np.asarray(self.caf_metas[0].skeleton, dtype=np.int64) - 1
            for b in range(B):
                a, i = port.decode(cif[b], 8, caf[b], 8, skel)
                counts[b] = len(a); out[b, :len(a)] = a; ids[b, :len(a)] = i
            return torch.from_numpy(out), torch.from_numpy(ids), torch.from_numpy(counts)

    class Model(torch.nn.Module):                # image i is a frame of grey level 10 i: the "network" returns that image's fields
        def __init__(self):
            super().__init__()
            self.head_metas = list(headmeta.cocokp_metas())
            self.seen = []
        def forward(self, x):
            idx = [int(round(float((x[b, 0, x.shape[2] // 2, x.shape[3] // 2] * 0.229 + 0.485) * 255.0) / 10.0)) for b in range(len(x))]
            self.seen.append(idx)
            return (torch.from_numpy(np.stack([fields[i][0] for i in idx])), torch.from_numpy(np.stack([fields[i][1] for i in idx])))

    decoder.factory = lambda metas: decoder.Multi([OracleCifCaf([metas[0]], [metas[1]])])
    predictor.Predictor.device, predictor.Predictor.batch_size, predictor.Predictor.long_edge = torch.device('cpu'), 4, 65
    frames = [np.full((49, 65, 3), 10 * i, dtype=np.uint8) for i in range(N)]

    def run(distributed):
        OracleCifCaf.distributed = distributed
        model = Model()
        pred = predictor.Predictor(model=model)
        gathers = []
        real = dist.all_gather
        dist.all_gather = lambda *a_, **k_: (gathers.append(1), real(*a_, **k_))[1]
        try:
            res = [[(a.data.copy(), a.joint_scales.copy()) for a in p] for p, _, _ in pred.numpy_images(frames)]
        finally:
            dist.all_gather = real
        return res, model.seen, len(gathers)

    alone, seen_alone, g0 = run(False)            # every rank on its own: whole batches, no collective
    assert g0 == 0 and seen_alone == [[0, 1, 2, 3], [4, 5, 6]], (g0, seen_alone)
    sharded, seen, g1 = run(None)                 # automatic: the process group has two ranks
    assert g1 == 2, 'ONE collective per batch (two batches), saw %%d' %% g1
    # batch of 4: two images per rank; batch of 3: two + one, the short shard padded with a repeated image
    want_seen = [[0, 1], [4, 5]] if rank == 0 else [[2, 3], [6, 6]]
    assert seen == want_seen, (rank, seen)
    assert len(sharded) == N and sum(len(p) for p in sharded) > N
    for i in range(N):
        assert len(sharded[i]) == len(alone[i]), i
        for (d1, s1), (d2, s2) in zip(sharded[i], alone[i]):
            assert np.array_equal(d1, d2) and np.array_equal(s1, s2), i
    # a failed image of the OTHER rank's shard raises here too (its flag travels with its count through the gather)
    class Failing(OracleCifCaf):
        def decode_heads(self, heads):
            out, ids, counts = OracleCifCaf.decode_heads(self, heads)
            if rank == 1:
                counts[0] = native.COUNT_FAILED
            return out, ids, counts
        class cpp_decoder:
            pool_overflowed = staticmethod(lambda: False)
    decoder.factory = lambda metas: decoder.Multi([Failing([metas[0]], [metas[1]])])
    Failing.distributed = None
    from openpifpaf_amd import _lib
    try:
        list(predictor.Predictor(model=Model()).numpy_images(frames[:4]))
        raise SystemExit('rank %%d: a failed image of the gathered batch passed' %% rank)
    except _lib.NativeError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
''') % ROOT


def test_predictor_shards_batches_over_two_gloo_ranks(tmp_path):
    script = tmp_path / 'product_worker.py'
    script.write_text(PRODUCT_WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert 'rank %d ok' % rank in out
