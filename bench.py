#!/usr/bin/env python
"""bench.py -- images/s end-to-end (backbone + CifCaf decode) on N MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by
``torch.distributed.run`` with one rank per GPU) prints ONE JSON line on rank 0.

A step = one pass of the inference path over one batch resident in HBM:
  1. backbone + CIF/CAF heads (ResNet-50 @ 641x641, random init, PyTorch-ROCm)
     on a synthetic image batch -- the real network work;
  2. the HIP CifCaf decode (CifHr -> CifSeeds -> CafScored -> association -> NMS)
     of COCO-shaped synthetic field tensors of exactly the heads' output shapes.
     A randomly initialised head emits structureless fields, so decode inputs are
     injected after the heads.  They are resident in HBM before
     the timed region; nothing is skipped or cached;
  3. final annotations: device -> pinned host copy; with N > 1 an RCCL all_gather
     of the fixed-size annotation blocks over xGMI (images shard one batch per GPU,
     no other collective).
The decode runs on a second HIP stream so that batch i's decode overlaps batch
i+1's backbone.

Besides the contract fields the line carries
  "roofline":     HBM roofline of the decode kernel that dominates the decode time,
                  from per-kernel HIP-event timings on the launch stream;
  "cpu_baseline": the reference CPU decoder (oracle/_ref, the reference's own C++)
                  timed on this box's host cores on a bounded sample of the same fields.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# MIOpen's find step otherwise also benchmarks its naive reference solver (~0.3 s per call)
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')

import torch  # noqa: E402  (after the MIOpen environment is set)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    p.add_argument('--backbone', default='resnet50')
    p.add_argument('--backbone-dtype', default='bf16', choices=('bf16', 'fp16', 'fp32'))
    p.add_argument('--long-edge', type=int, default=641)
    p.add_argument('--no-overlap', action='store_true', help='decode on the backbone stream')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline budget (rank 0, N=1)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--decode-only', action='store_true', help='skip the backbone (kernel work only)')
    p.add_argument('--graph', action='store_true',
                   help='decode-only: replay each decoder\'s call as a captured HIP graph')
    p.add_argument('--decode-streams', type=int, default=1,
                   help='decode-only: round-robin the batches over this many decoders, each with its own '
                        'stream and workspace (a batch-32 association occupies 32 of the 256 CUs)')
    p.add_argument('--profile-steps', type=int, default=5)
    p.add_argument('--force-complete', action='store_true',
                   help='decode like the reference\'s benchmark CLI (--force-complete-pose, thresholds 0)')
    p.add_argument('--fields', default='synthetic', choices=('synthetic', 'network'),
                   help='decode COCO-shaped synthetic fields injected after the heads (default), or the '
                        'random-init network\'s own all-active head outputs (adversarial case, reported separately)')
    p.add_argument('--dist-backend', default='nccl', choices=('nccl', 'gloo'),
                   help='nccl = RCCL over xGMI (default); gloo only to exercise the N>1 control flow on one GPU')
    p.add_argument('--share-device', action='store_true', help='testing: every rank uses cuda:0')
    return p.parse_args()


def algorithmic_bytes(B, F, A, H, W, stride, max_ann):
    """Per-launch algorithmic HBM bytes of each decode kernel."""
    hw = H * W
    rows, cols = (H - 1) * stride + 1, (W - 1) * stride + 1
    return {
        'cif_active_kernel': B * F * 4 * hw * 4,                 # reads conf,x,y,scale planes
        'cifhr_tile_kernel': B * F * rows * cols * 4,            # whole map; bench.py replaces it by the tiles actually written
        'cifseeds_fill_kernel': B * F * hw * 4,                  # reads the confidence plane
        'cifseeds_sort_kernel': 0,
        'cafscored_kernel': B * A * 7 * hw * 4,                  # reads the 7 used component planes
        'memset_occupancy': B * F * (rows // 2 + 1) * (cols // 2 + 1),
        'cifcaf_assoc_kernel': B * max_ann * F * 4 * 4,          # writes the annotations (lists are data dependent)
        'decode_path': B * (F * 5 * hw * 4 + A * 8 * hw * 4 + max_ann * F * 4 * 4),   # 6.24 MB/img
    }


def cpu_baseline(cifs, cafs, skeleton0, seconds, fc_kw=None):
    """The reference's own C++ decoder (oracle/_ref) on host cores: 1 thread and all cores."""
    from oracle import reference
    if not reference.available():
        from oracle import port
        port_params = port.default_params(**(fc_kw or {}))
        kind, decode = 'port', lambda c, f: port.decode(c, 8, f, 8, skeleton0, params=port_params)
        torch_ = None
    else:
        torch_ = reference.load()
        torch_.set_num_threads(1)
        reference.reset_statics()
        if fc_kw:
            from oracle import port
            reference.apply_params(port.default_params(**fc_kw))
        kind = 'reference'
        skel_t = torch_.as_tensor(skeleton0, dtype=torch_.int64)

        def decode(c, f):
            dec = torch_.classes.openpifpaf_decoder.CifCaf(int(c.shape[0]), skel_t)   # fresh instance per image
            return dec.call(torch_.from_numpy(c), 8, torch_.from_numpy(f), 8)
    n = len(cifs)
    decode(cifs[0], cafs[0])                       # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        decode(cifs[done % n], cafs[done % n])
        done += 1
        if done >= n and time.perf_counter() - t0 > seconds * 0.5:
            break
        if time.perf_counter() - t0 > seconds:
            break
    single = done / (time.perf_counter() - t0)

    # all host cores: the reference's --decoder-workers mechanism is a fork pool
    # (reference decoder/decoder.py:33-47,130-131); N = os.cpu_count().  Time-bounded.
    cores = os.cpu_count() or 1
    multi = None
    try:
        import multiprocessing as mp
        ctx = mp.get_context('fork')
        budget = max(3.0, 0.6 * seconds)

        def work(i, q, t_end):
            if torch_ is not None:
                torch_.set_num_threads(1)
            k = 0
            while time.perf_counter() < t_end or k == 0:
                decode(cifs[(i + k) % n], cafs[(i + k) % n])
                k += 1
            q.put((k, time.perf_counter()))
        q = ctx.Queue()
        t0 = time.perf_counter()
        t_end = t0 + budget
        procs = [ctx.Process(target=work, args=(i, q, t_end)) for i in range(cores)]
        for pr in procs:
            pr.start()
        results = [q.get() for _ in procs]
        for pr in procs:
            pr.join()
        multi = sum(k for k, _ in results) / (max(t for _, t in results) - t0)
    except Exception as e:   # pragma: no cover
        multi = None
        print('cpu_baseline: multi-process leg failed: %r' % (e,), file=sys.stderr)
    return {
        'value': round(single, 2), 'unit': 'images/s (decode only, 1 thread)', 'cores': 1, 'kind': kind,
        'all_cores_value': round(multi, 2) if multi else None, 'all_cores': cores,
        'sample': '%d decodes of the rank-0 batch fields (fresh decoder per image), single thread; '
                  'then %d forked single-thread workers for a fixed time budget' % (done, cores),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    # MIOpen "find" mode: benchmark the real solvers once per conv shape during warm-up.  The
    # immediate-mode heuristic occasionally falls back to naive_conv (~300 ms per call).
    torch.backends.cudnn.benchmark = True
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)  # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group('gloo')
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    from openpifpaf_amd import _lib, constants, distributed, headmeta, native, network, synth

    B = args.batch
    cif_meta, caf_meta = headmeta.cocokp_metas()
    skeleton0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
    dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.backbone_dtype]

    # ---- model + resident inputs
    model = None
    if not args.decode_only:
        model = network.factory(args.backbone, [cif_meta, caf_meta]).to(device)
        network.optimize_for_inference_(model)
        model = model.to(memory_format=torch.channels_last)
        if dtype != torch.float32:
            model = model.to(dtype)
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    images = torch.randn((B, 3, args.long_edge, args.long_edge), generator=g).to(device)
    images = images.contiguous(memory_format=torch.channels_last)
    if dtype != torch.float32:
        images = images.to(dtype)
    fh = (args.long_edge - 1) // 16 * 2 + 1            # 641 -> 41 -> 82 -> 81
    cifs_np, cafs_np = synth.synth_batch(B, seed0=rank * B, height=fh, width=fh)
    cif_syn = torch.from_numpy(cifs_np).to(device)
    caf_syn = torch.from_numpy(cafs_np).to(device)
    stride = cif_meta.stride

    fc_kw = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                 nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)     # reference decoder/cifcaf.py:180-185
    dec_params = _lib.default_params(**fc_kw) if args.force_complete else None
    dec = native.CifCaf(17, torch.from_numpy(skeleton0))
    n_streams = max(1, args.decode_streams) if args.decode_only else 1
    extra = [(native.CifCaf(17, torch.from_numpy(skeleton0)), torch.cuda.Stream(priority=-1))
             for _ in range(n_streams - 1)]
    step_no = [0]
    graphs = None
    if args.decode_only and args.graph:              # one captured decode per (decoder, stream)
        lanes = [(dec, torch.cuda.Stream(priority=-1))] + extra
        graphs = []
        for d, st in lanes:
            g, outs = d.capture(cif_syn, stride, caf_syn, stride, params=dec_params, stream=st)
            graphs.append((g, st, outs))
    K = 17
    host_out = torch.empty((B, dec.max_annotations, K, 4), dtype=torch.float32).pin_memory()
    host_counts = torch.empty((B,), dtype=torch.int32).pin_memory()
    main_stream = torch.cuda.current_stream()
    # high priority: the few decode workgroups slip in between the backbone's waves instead of queueing behind them
    dec_stream = main_stream if args.no_overlap else torch.cuda.Stream(priority=-1)

    shapes_checked = [False]

    def step():
        heads = None
        if model is not None:
            with torch.no_grad():
                heads = model(images)
            if not shapes_checked[0]:
                assert tuple(heads[0].shape) == tuple(cif_syn.shape), (heads[0].shape, cif_syn.shape)
                assert tuple(heads[1].shape) == tuple(caf_syn.shape), (heads[1].shape, caf_syn.shape)
                shapes_checked[0] = True
        step_no[0] += 1
        if graphs is not None:
            g, st, (out, ids, counts) = graphs[step_no[0] % n_streams]
            with torch.cuda.stream(st):
                g.replay()
                host_out.copy_(out, non_blocking=True)
                host_counts.copy_(counts, non_blocking=True)
            return out
        if n_streams > 1 and step_no[0] % n_streams:   # decode-only: this batch goes to one of the extra decoders
            d, st = extra[step_no[0] % n_streams - 1]
            with torch.cuda.stream(st):
                out, ids, counts = d.call_batch(cif_syn, stride, caf_syn, stride, params=dec_params)
                host_out.copy_(out, non_blocking=True)
                host_counts.copy_(counts, non_blocking=True)
            return out
        ev = torch.cuda.Event()
        ev.record(main_stream)
        with torch.cuda.stream(dec_stream):
            dec_stream.wait_event(ev)                  # decode of batch i follows its backbone
            if args.fields == 'network' and model is not None:
                out, ids, counts = dec.call_batch(heads[0], stride, heads[1], stride, params=dec_params)
            else:
                out, ids, counts = dec.call_batch(cif_syn, stride, caf_syn, stride, params=dec_params)
            if world > 1:                              # final annotations only, over xGMI (RCCL)
                if args.dist_backend == 'nccl':
                    distributed.gather_annotations(out, ids, counts)
                else:
                    distributed.gather_annotations(out.cpu(), ids.cpu(), counts.cpu())
            host_out.copy_(out, non_blocking=True)
            host_counts.copy_(counts, non_blocking=True)
        return out

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    t_setup = time.perf_counter()
    for _ in range(args.warmup):
        step()
    sync_all()
    if rank == 0:
        print('bench: warm-up (incl. MIOpen find) %.1f s' % (time.perf_counter() - t_setup), file=sys.stderr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_ann = int((host_counts & 0x3FFFFFFF).sum())           # OPA_COUNT_ROWS

    # ---- rank 0: roofline leg (per-kernel HIP-event timings) and CPU baseline
    result = None
    if rank == 0:
        per_kernel = {}
        with torch.cuda.stream(dec_stream):
            for _ in range(max(1, args.profile_steps)):
                _lib.profile_begin(native._stream())
                dec.call_batch(cif_syn, stride, caf_syn, stride, params=dec_params)
                for name, ms in _lib.profile_end():
                    per_kernel.setdefault(name, []).append(ms)
        avg_ms = {k: float(np.mean(v)) for k, v in per_kernel.items()}
        alg = algorithmic_bytes(B, 17, 19, fh, fh, stride, dec.max_annotations)
        # the tile kernel writes only the 32x64 tiles this call's or the previous call's cells reach
        # (lazy clear): its algorithmic bytes are those tiles, counted from the bitmaps in the workspace
        bitmaps = dec.workspace_view('tile_bitmaps', torch.int32).cpu().numpy().view(np.uint32)
        words = ((((fh - 1) * stride + 1 + 63) // 64) * (((fh - 1) * stride + 1 + 31) // 32) + 31) // 32
        tiles_written = int(np.unpackbits(bitmaps[:B * 17 * words].view(np.uint8)).sum())
        alg['cifhr_tile_kernel'] = tiles_written * 32 * 64 * 4
        decode_ms = sum(avg_ms.values())
        dominant = max(avg_ms, key=avg_ms.get)
        dom_bytes = alg.get(dominant, 0)
        achieved = dom_bytes / (avg_ms[dominant] * 1e-3) / 1e9 if avg_ms[dominant] > 0 else 0.0
        # HBM bytes per launch from the committed rocprofv3 PMC passes (same command, same batch);
        # bench.py cannot collect PMC counters itself
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r1', 'pmc_traffic.json')))
            if pmc.get('batch') == B:
                traffic = pmc['kernels'].get(dominant, {}).get('hbm_bytes')
        except (OSError, ValueError):
            pass
        roofline = {
            'bound': 'hbm', 'kernel': dominant, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 5), 'traffic': traffic,
            'traffic_source': 'profiles/r1/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, x2 read correction)',
            'avg_launch_ms': round(avg_ms[dominant], 4), 'algorithmic_bytes_per_launch': dom_bytes,
            'kernels': {k: {'ms': round(v, 4),
                            'GBps': round(alg.get(k, 0) / (v * 1e-3) / 1e9, 1) if v > 0 else None}
                        for k, v in avg_ms.items()},
            'decode_path': {'ms_per_batch': round(decode_ms, 4),
                            'images_per_s': round(B / (decode_ms * 1e-3), 1),
                            'GBps': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9, 2),
                            'frac': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cifs_np, cafs_np, skeleton0, args.cpu_seconds, fc_kw if args.force_complete else None)
        value = world * B * args.steps / elapsed
        result = {
            'metric': ('images/sec end-to-end (backbone+CifCaf decode), resnet50 641px' if not args.decode_only else
                       'images/sec DECODE ONLY (diagnostic: backbone skipped, not the headline metric)'),
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'configs[1] scaled to a batch: %s %dx%d, batch %d per GPU, COCO-17 CIF/CAF fields '
                            '[%d,17,5,%d,%d]+[%d,19,8,%d,%d]' % (args.backbone, args.long_edge, args.long_edge,
                                                                 B, B, fh, fh, B, fh, fh),
                'backbone': 'none (decode only)' if model is None else args.backbone,
                'backbone_dtype': args.backbone_dtype, 'decode_dtype': 'f32 (+f64 where the reference uses double)',
                'global_batch': world * B,
                'fields': ('COCO-shaped synthetic fields injected after the heads (people per image cycle %s)'
                           % (list(synth.PEOPLE_CYCLE),)) if args.fields == 'synthetic' else
                          'the random-init network\'s own head outputs (all-active adversarial case)',
                'parallelism': 'images sharded one batch per GPU (dp%d); RCCL all_gather of annotations' % world
                               if world > 1 else 'single GPU',
                'decode_overlapped_on_second_stream': not args.no_overlap,
                'decode_streams': n_streams, 'hip_graph': graphs is not None,
                'force_complete_pose': bool(args.force_complete),
                'annotations_per_batch': n_ann,
            },
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        if cpu is not None:
            result['decode_vs_cpu_1thread'] = round(roofline['decode_path']['images_per_s'] / cpu['value'], 1)
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
