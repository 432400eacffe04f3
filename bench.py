#!/usr/bin/env python
"""bench.py -- images/s end-to-end (backbone + CifCaf decode) on N MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by
``torch.distributed.run`` with one rank per GPU) prints ONE JSON line on rank 0.

A step = one pass of the inference path over one batch resident in HBM:
  1. backbone + CIF/CAF heads (random init, PyTorch-ROCm + this library's producer-side kernels) on a
     synthetic image batch -- the real network work;
  2. the HIP CifCaf decode (CifHr -> CifSeeds -> CafScored -> association -> NMS) of COCO-shaped
     synthetic field tensors of exactly the heads' output shapes.  A randomly initialised head emits
     structureless fields, so decode inputs are injected after the heads.  They are
     resident in HBM before the timed region; nothing is skipped or cached;
  3. final annotations: device -> pinned host copy; with N > 1 ONE RCCL all_gather of the packed
     annotation blocks over xGMI (images shard one batch per GPU, no other collective).
The decode runs on a second HIP stream so that batch i's decode overlaps batch i+1's backbone.

``--config`` selects the BASELINE.json configuration (default 2 = configs[1] scaled to a batch, the one
the metric is quoted on): 2 resnet50 COCO-17, 3 shufflenetv2k16 COCO-17, 4 shufflenetv2k30 wholebody
(133 keypoints / 160 bones, batch 16).

LIKE FOR LIKE.  The reference runs its network in float32 (``predictor.py:33-41``), so ``value`` is the
end-to-end rate with a FLOAT32 backbone; the bfloat16-backbone rate is reported beside it
(``bf16_backbone``).  ``vs_baseline`` divides ``value`` by the reference's own data flow measured in the
same run with the same backbone precision: backbone on the MI355X -> ``.cpu()`` of the head fields
(``decoder/decoder.py:96-100``) -> the reference's C++ CifCaf decoder (oracle/_ref) on one host thread,
one decoder instance reused across images like ``decoder/cifcaf.py:119`` does.  The all-host-cores
variant (``--decoder-workers``, a fork pool) is in ``reference_pipeline`` too.

Besides the contract fields the line carries
  "roofline":     HBM roofline of the decode kernel that dominates the decode time: SURVEY 8d bytes per
                  launch / its HIP-event time, with the kernel's own compulsory bytes beside it;
  "cpu_baseline": the reference CPU decoder timed on this box's host cores on a bounded sample of the
                  same fields (reused instance, fresh instance, all cores).
"""
import argparse
import hashlib
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
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}     # dense matrix peaks, MI355X_MICROARCH.md

CONFIGS = {
    2: dict(name='configs[1] scaled to a batch', backbone='resnet50', batch=32, wholebody=False),
    3: dict(name='configs[2]', backbone='shufflenetv2k16', batch=32, wholebody=False),
    4: dict(name='configs[3]', backbone='shufflenetv2k30', batch=16, wholebody=True),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    p.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: the config\'s)')
    p.add_argument('--backbone', default=None, help='override the config\'s backbone')
    p.add_argument('--backbone-dtype', default='fp32', choices=('bf16', 'fp16', 'fp32'),
                   help='precision of the HEADLINE leg (fp32 = the reference\'s)')
    p.add_argument('--no-bf16-leg', action='store_true', help='skip the second, bfloat16-backbone leg')
    p.add_argument('--long-edge', type=int, default=641)
    p.add_argument('--no-overlap', action='store_true', help='decode on the backbone stream')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline budget (rank 0, N=1)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--decode-only', action='store_true', help='skip the backbone (kernel work only; diagnostic)')
    p.add_argument('--graph', action='store_true',
                   help='decode-only: replay each decoder\'s call as a captured HIP graph')
    p.add_argument('--decode-streams', type=int, default=1,
                   help='decode-only: round-robin the batches over this many decoders, each with its own '
                        'stream and workspace')
    p.add_argument('--profile-steps', type=int, default=5)
    p.add_argument('--force-complete', action='store_true',
                   help='decode like the reference\'s benchmark CLI (--force-complete-pose, thresholds 0)')
    p.add_argument('--fields', default='synthetic', choices=('synthetic', 'network'),
                   help='decode COCO-shaped synthetic fields injected after the heads (default), or the '
                        'random-init network\'s own all-active head outputs (adversarial case, reported separately)')
    p.add_argument('--dist-backend', default='nccl', choices=('nccl', 'gloo'),
                   help='nccl = RCCL over xGMI (default); gloo only to exercise the N>1 control flow on one GPU')
    p.add_argument('--share-device', action='store_true', help='testing: every rank uses cuda:0')
    p.add_argument('--dump-annotations', default=None,
                   help='testing: rank 0 writes the gathered annotations of the last step to this .npz')
    return p.parse_args()


def algorithmic_bytes(B, F, A, H, W, stride, max_ann, K=None):
    """Per-launch compulsory HBM bytes of each decode kernel."""
    hw = H * W
    K = K or F
    rows, cols = (H - 1) * stride + 1, (W - 1) * stride + 1
    return {
        'cif_active_kernel': B * F * 4 * hw * 4,                 # reads conf,x,y,scale planes
        'cifhr_tile_kernel': B * F * rows * cols * 4,            # whole map; replaced below by the tiles actually written
        'cifseeds_fill_kernel': B * F * hw * 4,                  # reads the confidence plane
        'cifseeds_sort_kernel': 0,
        'cafscored_kernel': B * A * 7 * hw * 4,                  # reads the 7 used component planes
        'cifcaf_assoc_kernel': B * max_ann * K * 4 * 4,          # writes the annotations (lists are data dependent)
        'decode_path': B * (F * 5 * hw * 4 + A * 8 * hw * 4 + max_ann * K * 4 * 4),
    }


def kernel_source_hash():
    """Identifies the kernels a PMC traffic file was measured on (profiles/r2/pmc_traffic.json)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'openpifpaf_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.hpp')):
            h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def reference_decoder(skeleton0, n_keypoints, fc_kw=None):
    """-> (kind, make_decode) where make_decode(reuse) returns decode(cif, caf) on the host: the reference's own
    C++ decoder (oracle/_ref) when it is present, the restatement otherwise."""
    from oracle import reference
    if not reference.available():
        from oracle import port
        port_params = port.default_params(**(fc_kw or {}))
        return 'port', lambda reuse: (lambda c, f: port.decode(c, 8, f, 8, skeleton0, params=port_params)), None
    torch_ = reference.load()
    torch_.set_num_threads(1)
    reference.reset_statics()
    if fc_kw:
        from oracle import port
        reference.apply_params(port.default_params(**fc_kw))
    skel_t = torch_.as_tensor(skeleton0, dtype=torch_.int64)

    def make(reuse):
        shared = torch_.classes.openpifpaf_decoder.CifCaf(int(n_keypoints), skel_t) if reuse else None

        def decode(c, f):
            dec = shared if shared is not None else torch_.classes.openpifpaf_decoder.CifCaf(int(n_keypoints), skel_t)
            return dec.call(torch_.from_numpy(c), 8, torch_.from_numpy(f), 8)
        return decode
    return 'reference', make, torch_


def time_loop(decode, cifs, cafs, seconds, min_items):
    n = len(cifs)
    decode(cifs[0], cafs[0])                       # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        decode(cifs[done % n], cafs[done % n])
        done += 1
        dt = time.perf_counter() - t0
        if (done >= min_items and dt > seconds * 0.5) or dt > seconds:
            break
    return done / (time.perf_counter() - t0), done


def cpu_baseline(cifs, cafs, skeleton0, n_keypoints, seconds, fc_kw=None):
    """The reference's own C++ decoder on host cores: one thread with a reused and with a fresh decoder
    instance per image, and a fork pool on all cores."""
    kind, make, torch_ = reference_decoder(skeleton0, n_keypoints, fc_kw)
    n = len(cifs)
    reused, n_reused = time_loop(make(True), cifs, cafs, seconds * 0.3, n)
    fresh, n_fresh = time_loop(make(False), cifs, cafs, seconds * 0.3, min(n, 8))

    # all host cores: the reference's --decoder-workers mechanism is a fork pool
    # (reference decoder/decoder.py:33-47,130-131); N = os.cpu_count().  Time-bounded.
    cores = os.cpu_count() or 1
    multi = None
    try:
        import multiprocessing as mp
        ctx = mp.get_context('fork')
        budget = max(3.0, 0.4 * seconds)
        decode = make(True)

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
        'value': round(reused, 2), 'unit': 'images/s (decode only, 1 thread, decoder instance reused)', 'cores': 1,
        'kind': kind, 'fresh_instance_value': round(fresh, 2),
        'all_cores_value': round(multi, 2) if multi else None, 'all_cores': cores,
        'sample': '%d decodes of the rank-0 batch fields with one reused decoder instance (what the reference\'s '
                  'Decoder does), %d with a fresh instance per image (the parity definition: revision drift), '
                  'then %d forked single-thread workers for a fixed time budget' % (n_reused, n_fresh, cores),
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

    cfg = CONFIGS[args.config]
    backbone = args.backbone or cfg['backbone']
    B = args.batch or cfg['batch']
    if cfg['wholebody']:
        wb = constants.wholebody()
        cif_meta, caf_meta = headmeta.wholebody_metas()
        skeleton1, pose, people = wb['skeleton'], wb['standing_pose'], (1, 3, 6, 10)
    else:
        cif_meta, caf_meta = headmeta.cocokp_metas()
        skeleton1, pose, people = constants.COCO_PERSON_SKELETON, None, synth.PEOPLE_CYCLE
    skeleton0 = np.asarray(skeleton1, dtype=np.int64) - 1
    K, A = cif_meta.n_fields, caf_meta.n_fields
    TORCH_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}

    # ---- resident inputs
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    images32 = torch.randn((B, 3, args.long_edge, args.long_edge), generator=g).to(device)
    images32 = images32.contiguous(memory_format=torch.channels_last)
    fh = (args.long_edge - 1) // 16 * 2 + 1            # 641 -> 41 -> 82 -> 81
    cifs_np, cafs_np = synth.synth_batch(B, seed0=rank * B, height=fh, width=fh, people=people, pose=pose,
                                         skeleton=skeleton1 if cfg['wholebody'] else None)
    cif_syn = torch.from_numpy(cifs_np).to(device)
    caf_syn = torch.from_numpy(cafs_np).to(device)
    stride = cif_meta.stride

    fc_kw = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
                 nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)     # reference decoder/cifcaf.py:180-185
    dec_params = _lib.default_params(**fc_kw) if args.force_complete else None
    dec = native.CifCaf(K, torch.from_numpy(skeleton0))
    n_streams = max(1, args.decode_streams) if args.decode_only else 1
    extra = [(native.CifCaf(K, torch.from_numpy(skeleton0)), torch.cuda.Stream(priority=-1))
             for _ in range(n_streams - 1)]
    graphs = None
    if args.decode_only and args.graph:              # one captured decode per (decoder, stream)
        lanes = [(dec, torch.cuda.Stream(priority=-1))] + extra
        graphs = []
        for d, st in lanes:
            gr, outs = d.capture(cif_syn, stride, caf_syn, stride, params=dec_params, stream=st)
            graphs.append((gr, st, outs))
    host_out = torch.empty((B, dec.max_annotations, K, 4), dtype=torch.float32).pin_memory()
    host_counts = torch.empty((B,), dtype=torch.int32).pin_memory()
    main_stream = torch.cuda.current_stream()
    # high priority: the few decode workgroups slip in between the backbone's waves instead of queueing behind them
    dec_stream = main_stream if args.no_overlap else torch.cuda.Stream(priority=-1)
    gathered = [None]

    def build_model(dtype_name):
        model = network.factory(backbone, [cif_meta, caf_meta]).to(device)
        network.optimize_for_inference_(model)
        model = model.to(memory_format=torch.channels_last)
        if dtype_name != 'fp32':
            model = model.to(TORCH_DTYPE[dtype_name])
        return model

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    def run_leg(dtype_name, steps, warmup):
        """W warm-up steps, then EXACTLY `steps` timed steps between barrier + synchronize; -> (max-over-ranks
        seconds, the model)."""
        model = None if args.decode_only else build_model(dtype_name)
        images = images32 if dtype_name == 'fp32' else images32.to(TORCH_DTYPE[dtype_name])
        step_no = [0]
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
                gr, st, (out, ids, counts) = graphs[step_no[0] % n_streams]
                with torch.cuda.stream(st):
                    gr.replay()
                    host_out.copy_(out, non_blocking=True)
                    host_counts.copy_(counts, non_blocking=True)
                return
            if n_streams > 1 and step_no[0] % n_streams:   # decode-only: this batch goes to one of the extra decoders
                d, st = extra[step_no[0] % n_streams - 1]
                with torch.cuda.stream(st):
                    out, ids, counts = d.call_batch(cif_syn, stride, caf_syn, stride, params=dec_params)
                    host_out.copy_(out, non_blocking=True)
                    host_counts.copy_(counts, non_blocking=True)
                return
            ev = torch.cuda.Event()
            ev.record(main_stream)
            with torch.cuda.stream(dec_stream):
                dec_stream.wait_event(ev)                  # decode of batch i follows its backbone
                if args.fields == 'network' and model is not None:
                    out, ids, counts = dec.call_batch(heads[0], stride, heads[1], stride, params=dec_params)
                else:
                    out, ids, counts = dec.call_batch(cif_syn, stride, caf_syn, stride, params=dec_params)
                if world > 1:                              # final annotations only, ONE collective (RCCL over xGMI)
                    if args.dist_backend == 'nccl':
                        gathered[0] = distributed.gather_annotations(out, ids, counts)
                    else:
                        gathered[0] = distributed.gather_annotations(out.cpu(), ids.cpu(), counts.cpu())
                host_out.copy_(out, non_blocking=True)
                host_counts.copy_(counts, non_blocking=True)

        t_setup = time.perf_counter()
        for _ in range(warmup):
            step()
        sync_all()
        if rank == 0:
            print('bench: %s leg warm-up (incl. MIOpen find) %.1f s' % (dtype_name, time.perf_counter() - t_setup),
                  file=sys.stderr)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.dist_backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, model, images

    def backbone_only_ms(model, images, reps=5):
        """Network alone (backbone + heads) per batch, and the host copy of its field tensors the reference
        makes before decoding (decoder/decoder.py:96-100)."""
        with torch.no_grad():
            heads = model(images)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(reps):
                heads = model(images)
        torch.cuda.synchronize(device)
        nn_ms = (time.perf_counter() - t0) / reps * 1e3
        [h.cpu() for h in heads]                        # first copy allocates
        t0 = time.perf_counter()
        for _ in range(3):
            host = [h.cpu() for h in heads]
        d2h_ms = (time.perf_counter() - t0) / 3 * 1e3
        return nn_ms, d2h_ms, sum(h.numel() * h.element_size() for h in host)

    # ---- the headline leg (reference precision), then the bfloat16 one
    legs = {}
    primary = args.backbone_dtype
    elapsed, model, images = run_leg(primary, args.steps, args.warmup)
    legs[primary] = dict(elapsed=elapsed, value=world * B * args.steps / elapsed)
    n_ann = int((host_counts & 0x0FFFFFFF).sum())           # OPA_COUNT_ROWS
    if args.dump_annotations and rank == 0:
        a_, i_, c_ = gathered[0] if gathered[0] is not None else (host_out, None, host_counts)
        np.savez(args.dump_annotations, annotations=a_.cpu().numpy(), counts=c_.cpu().numpy())
    nn = {}
    if model is not None and rank == 0 and world == 1:
        nn[primary] = backbone_only_ms(model, images)
    del model
    if not args.decode_only and not args.no_bf16_leg and primary != 'bf16':
        elapsed2, model2, images2 = run_leg('bf16', args.steps, args.warmup)
        legs['bf16'] = dict(elapsed=elapsed2, value=world * B * args.steps / elapsed2)
        if rank == 0 and world == 1:
            nn['bf16'] = backbone_only_ms(model2, images2)
        del model2

    # ---- rank 0: roofline leg (per-kernel HIP-event timings) and CPU legs
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
        alg = algorithmic_bytes(B, K, A, fh, fh, stride, dec.max_annotations)
        # the tile kernel writes only the 32x64 tiles this call's or the previous call's cells reach
        # (lazy clear): its compulsory bytes are those tiles, counted from the bitmaps in the workspace
        bitmaps = dec.workspace_view('tile_bitmaps', torch.int32).cpu().numpy().view(np.uint32)
        words = ((((fh - 1) * stride + 1 + 63) // 64) * (((fh - 1) * stride + 1 + 31) // 32) + 31) // 32
        tiles_written = int(np.unpackbits(bitmaps[:B * K * words].view(np.uint8)).sum())
        alg['cifhr_tile_kernel'] = tiles_written * 32 * 64 * 4
        decode_ms = sum(avg_ms.values())
        dominant = max(avg_ms, key=avg_ms.get)
        dom_ms = avg_ms[dominant]
        achieved = alg['decode_path'] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0       # SURVEY 8d bytes per launch
        own = alg.get(dominant, 0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch: rocprofv3 PMC passes cannot run inside bench.py; tools/collect_profiles.sh writes them
        # to profiles/r2/pmc_traffic.json stamped with the hash of the kernel sources they were measured on
        traffic, traffic_note = None, 'no PMC file for these kernel sources (run tools/collect_profiles.sh)'
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r2', 'pmc_traffic.json')))
            if pmc.get('batch') == B and pmc.get('config') == args.config and \
                    pmc.get('kernel_source_hash') == kernel_source_hash():
                traffic = pmc['kernels'].get(dominant, {}).get('hbm_bytes')
                traffic_note = 'profiles/r2/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, x2 read correction)'
        except (OSError, ValueError):
            pass
        roofline = {
            'bound': 'hbm', 'kernel': dominant, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 5), 'traffic': traffic, 'traffic_source': traffic_note,
            'avg_launch_ms': round(dom_ms, 4), 'algorithmic_bytes_per_launch': alg['decode_path'],
            'algorithmic_bytes_definition': 'SURVEY 8d: CIF + CAF + annotations of one image x images per launch',
            'own_bytes': {'bytes_per_launch': alg.get(dominant, 0), 'GBps': round(own, 2),
                          'frac': round(own / HBM_PEAK_GBPS, 6),
                          'note': 'the kernel\'s own compulsory bytes (the association kernel only writes the '
                                  'annotations: a latency-bound dependency chain, not a bandwidth problem)'},
            'kernels': {k: {'ms': round(v, 4),
                            'GBps': round(alg.get(k, 0) / (v * 1e-3) / 1e9, 1) if v > 0 else None}
                        for k, v in avg_ms.items()},
            'decode_path': {'ms_per_batch': round(decode_ms, 4),
                            'images_per_s': round(B / (decode_ms * 1e-3), 1),
                            'GBps': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9, 2),
                            'frac': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
        }
        if not args.decode_only and not cfg['wholebody'] and backbone == 'resnet50' and nn:
            gflop = 274.0 * B                             # 137 GMAC per 641x641 image
            roofline['backbone_mfma'] = {
                d: {'ms_per_batch': round(nn[d][0], 2), 'TFLOPs': round(gflop / nn[d][0], 1),
                    'frac_of_dense_peak': round(gflop / nn[d][0] / MFMA_PEAK_TFLOPS[d], 3)} for d in nn}
        cpu = None
        ref_pipe = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cifs_np, cafs_np, skeleton0, K, args.cpu_seconds, fc_kw if args.force_complete else None)
            if nn:
                # the reference's data flow, per batch: network on the GPU, head fields to the host, CPU decode
                ref_pipe = {}
                for d, (nn_ms, d2h_ms, field_bytes) in nn.items():
                    one = B / ((nn_ms + d2h_ms) * 1e-3 + B / cpu['value'])
                    allc = B / ((nn_ms + d2h_ms) * 1e-3 + B / cpu['all_cores_value']) if cpu.get('all_cores_value') else None
                    ref_pipe[d] = {
                        'network_ms_per_batch': round(nn_ms, 2), 'fields_to_host_ms_per_batch': round(d2h_ms, 2),
                        'field_bytes_per_batch': field_bytes,
                        'cpu_decode_ms_per_batch_1thread': round(B / cpu['value'] * 1e3, 1),
                        'images_per_s_1thread': round(one, 1),
                        'images_per_s_all_cores': round(allc, 1) if allc else None, 'cores': cpu['all_cores'],
                    }
        value = legs[primary]['value']
        result = {
            'metric': ('images/sec end-to-end (backbone+CifCaf decode), %s 641px' % backbone if not args.decode_only else
                       'images/sec DECODE ONLY (diagnostic: backbone skipped, not the headline metric)'),
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(legs[primary]['elapsed'] / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': round(value / ref_pipe[primary]['images_per_s_1thread'], 2) if ref_pipe else None,
            'vs_baseline_definition': 'value / the reference data flow measured in this run at the same backbone '
                                      'precision: network on the MI355X -> .cpu() of the fields -> reference C++ '
                                      'decoder, 1 host thread (reference_pipeline)',
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'BASELINE %s: %s %dx%d, batch %d per GPU, CIF/CAF fields [%d,%d,5,%d,%d]+[%d,%d,8,%d,%d]'
                            % (cfg['name'], backbone, args.long_edge, args.long_edge, B, B, K, fh, fh, B, A, fh, fh),
                'backbone': 'none (decode only)' if args.decode_only else backbone,
                'backbone_dtype': primary, 'decode_dtype': 'f32 (+f64 where the reference uses double)',
                'global_batch': world * B,
                'fields': ('COCO-shaped synthetic fields injected after the heads (people per image cycle %s)'
                           % (list(people),)) if args.fields == 'synthetic' else
                          'the random-init network\'s own head outputs (all-active adversarial case)',
                'parallelism': 'images sharded one batch per GPU (dp%d); one RCCL all_gather of the packed annotations' % world
                               if world > 1 else 'single GPU',
                'decode_overlapped_on_second_stream': not args.no_overlap,
                'decode_streams': n_streams, 'hip_graph': graphs is not None,
                'force_complete_pose': bool(args.force_complete),
                'annotations_per_batch': n_ann,
            },
            'roofline': roofline,
            'cpu_baseline': cpu,
            'reference_pipeline': ref_pipe,
        }
        if 'bf16' in legs and primary != 'bf16':
            v16 = legs['bf16']['value']
            result['bf16_backbone'] = {
                'value': round(v16, 2), 'ms_per_step': round(legs['bf16']['elapsed'] / args.steps * 1e3, 3),
                'vs_baseline_same_precision': round(v16 / ref_pipe['bf16']['images_per_s_1thread'], 2) if ref_pipe else None,
                'vs_reference_fp32_pipeline': round(v16 / ref_pipe[primary]['images_per_s_1thread'], 2)
                if ref_pipe and primary in ref_pipe else None,
                'note': 'same step with the network in bfloat16 (reduced precision relative to the reference; not the headline)',
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
