#!/usr/bin/env python
"""bench.py -- images/s end-to-end (backbone + CifCaf decode) on N MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by
``torch.distributed.run`` with one rank per GPU) prints ONE JSON line on rank 0.

A step = one pass of the inference path over one batch resident in HBM:
  1. backbone + CIF/CAF heads (random init, PyTorch-ROCm + this library's producer-side kernels) on a
     synthetic image batch -- the real network work;
  2. the HIP CifCaf decode (CifHr -> CifSeeds -> CafScored -> association -> NMS) of COCO-shaped
     synthetic field tensors of exactly the heads' output shapes.  A randomly initialised head emits
     structureless fields, so decode inputs are injected after the heads.  TWO different field
     batches alternate step by step (the lazy tile clear and the caches see changing input); both are resident
     in HBM before the timed region; nothing is skipped or cached;
  3. final annotations: device -> pinned host copy; with N > 1 ONE RCCL all_gather of the packed
     annotation blocks over xGMI (images shard one batch per GPU, no other collective).
The decode runs on a second HIP stream so that batch i's decode overlaps batch i+1's backbone.

``--config`` selects the BASELINE.json configuration (default 2 = configs[1] scaled to a batch, the one
the metric is quoted on): 2 resnet50 COCO-17, 3 shufflenetv2k16 COCO-17, 4 shufflenetv2k30 wholebody
(133 keypoints / 160 bones, batch 16).

LIKE FOR LIKE.  The reference runs its network in float32 (``predictor.py:33-41``), so ``value`` is the
end-to-end rate with a FLOAT32 backbone; the bfloat16-backbone rate is reported beside it
(``bf16_backbone``; its decode inputs are the same fields rounded to bfloat16, what a bf16 head emits, and its
parity stamp carries the seed-tie mismatch rate against the real reference).  ``vs_baseline`` divides ``value``
by the reference's own data flow RUN in the same process at the same backbone precision: backbone on the
MI355X -> ``.cpu()`` of the head fields (``decoder/decoder.py:96-100``) -> the reference's C++ CifCaf decoder
(oracle/_ref) on one host thread, one decoder instance reused across images like ``decoder/cifcaf.py:119``.

The default single-GPU run (no ``--config``) also carries, under ``"configs"``, short legs of the other BASELINE
configurations and settings, each with its own roofline / cpu_baseline / parity: ``config3`` (shufflenetv2k16
batch 32), ``config4`` (shufflenetv2k30 wholebody batch 16), ``force_complete`` (the reference benchmark CLI's
decoder setting, ``benchmark.py:77-79``), ``batch1`` (the literal configs[1]: resnet50 641x641 batch 1 latency,
eager and as one HIP graph, the reference flow at batch 1 beside it) and ``decode_two_in_flight``.

Besides the contract fields the line carries
  "roofline":     HBM roofline of the decode kernel that dominates the decode time: SURVEY 8d bytes per
                  launch / its HIP-event time, the kernel's own compulsory bytes beside it, and ``traffic`` =
                  PMC HBM bytes of the whole decode path per launch (profiles/r3/pmc_traffic.json, hash-stamped);
  "cpu_baseline": the reference CPU decoder timed on this box's host cores on a bounded sample of the
                  same fields (reused instance, fresh instance, all cores);
  "parity":       what the timed decode produces, compared image by image with the reference decoder OUTSIDE
                  the timed region.
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
# decode lanes want a hardware queue each (the runtime's default is four for all streams of the process)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402  (after the MIOpen environment is set)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}     # dense matrix peaks, MI355X_MICROARCH.md
TORCH_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}
PMC_ROUNDS = ('r6', 'r5', 'r4', 'r3')
VARIANT_SEED = 100_000   # field batch v of rank r is synth_batch(B, seed0 = v * VARIANT_SEED + r * B)
TOL = 1e-4               # BASELINE.json north_star: keypoint coordinates / scores within 1e-4

CONFIGS = {
    2: dict(name='configs[1] scaled to a batch', backbone='resnet50', batch=32, wholebody=False),
    3: dict(name='configs[2]', backbone='shufflenetv2k16', batch=32, wholebody=False),
    4: dict(name='configs[3]', backbone='shufflenetv2k30', batch=16, wholebody=True),
}
FC_KW = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
             nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)     # reference decoder/cifcaf.py:180-185


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--config', type=int, default=None, choices=sorted(CONFIGS),
                   help='one BASELINE configuration only (default: 2 as the headline plus short legs of the others)')
    p.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: the config\'s)')
    p.add_argument('--backbone', default=None, help='override the config\'s backbone')
    p.add_argument('--backbone-dtype', default='fp32', choices=('bf16', 'fp16', 'fp32'),
                   help='precision of the HEADLINE leg (fp32 = the reference\'s)')
    p.add_argument('--no-bf16-leg', action='store_true', help='skip the second, bfloat16-backbone leg')
    p.add_argument('--no-extras', action='store_true', help='skip the short legs of the other configurations')
    p.add_argument('--extra-steps', type=int, default=5)
    p.add_argument('--long-edge', type=int, default=641)
    p.add_argument('--no-overlap', action='store_true', help='decode on the backbone stream')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline budget (rank 0, N=1)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-parity', action='store_true', help='skip the parity stamp (it runs outside the timed region)')
    p.add_argument('--decode-only', action='store_true', help='skip the backbone (kernel work only; diagnostic)')
    p.add_argument('--graph', action='store_true',
                   help='decode-only: replay each decoder\'s call as a captured HIP graph (one field batch)')
    p.add_argument('--decode-streams', type=int, default=1,
                   help='decode-only: round-robin the batches over this many decoders, each with its own '
                        'stream and workspace')
    p.add_argument('--profile-steps', type=int, default=6)
    p.add_argument('--force-complete', action='store_true',
                   help='decode like the reference\'s benchmark CLI (--force-complete-pose, thresholds 0)')
    p.add_argument('--fields', default='synthetic', choices=('synthetic', 'network'),
                   help='decode COCO-shaped synthetic fields injected after the heads (default), or the '
                        'random-init network\'s own all-active head outputs (adversarial case, reported separately)')
    p.add_argument('--single-batch', action='store_true', help='do not alternate field batches (round-2 behaviour)')
    p.add_argument('--dist-backend', default='nccl', choices=('nccl', 'gloo'),
                   help='nccl = RCCL over xGMI (default); gloo only to exercise the N>1 control flow on one GPU')
    p.add_argument('--share-device', action='store_true', help='testing: every rank uses cuda:0')
    p.add_argument('--watchdog-seconds', type=float, default=600.0,
                   help='print what exists and exit when the run has not finished by then (a default run takes < 3 min)')
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
        'cifhr_tile_kernel': B * F * rows * cols * 4,            # whole map; replaced by the tiles actually written
        'cifseeds_fill_kernel': B * F * hw * 4,                  # reads the confidence plane
        'cifseeds_sort_kernel': 0,
        'cafscored_kernel': B * A * 7 * hw * 4,                  # reads the 7 used component planes
        'sort_cafscored_kernel': B * A * 7 * hw * 4,             # the decode's fused launch: seed sort (LDS resident) + cafscored
        'cifcaf_assoc_kernel': B * max_ann * K * 4 * 4,          # writes the annotations (lists are data dependent)
        'cifcaf_fc_kernel': B * max_ann * K * 4 * 4,
        'decode_path': B * (F * 5 * hw * 4 + A * 8 * hw * 4 + max_ann * K * 4 * 4),
    }


# producer-side kernels (the network's convolutions and epilogues): not launched by a decode, so an edit there does not
# invalidate a traffic file measured on the decode kernels
PRODUCER_SOURCES = ('dwconv.hip', 'epilogue.hip', 'gemm_epilogue.hip', 'gemm_f32.hip', 'gemm_f32x3.hip', 'head.hip', 'winograd.hip')


def _conv1x1_text(primary):
    """What the float32 1x1 convolutions ran on (openpifpaf_amd.fused: the choice table, after the legs have run)."""
    if primary != 'fp32':
        return 'bf16 MFMA GEMM, fused epilogue'
    try:
        from openpifpaf_amd import fused
        ch = [v for k, v in fused.choices().items() if k[0] == 'torch.float32']
        n3 = sum(1 for v in ch if v == 'gemm3')
        if n3 and fused.X3_TERMS:
            return ('f32 in/out; %d/%d shapes as 3 exact bf16 pieces x %d of 9 products on the bf16 MFMA, f32 accumulate '
                    '(error vs f64 < f32 MFMA\'s: f32_check); rest f32 MFMA' % (n3, len(ch), fused.X3_TERMS))
    except Exception:          # noqa: BLE001
        pass
    return 'f32 MFMA GEMM, fused epilogue'


def f32_numerics_check():
    """Outside every timed region: the float32 1x1 convolution of ResNet layer 3 (1024 -> 256 channels, 81x81, two images) by the
    kernels the headline may use -- split bf16 operands (csrc/gemm_f32x3.hip) and the float32 MFMA (csrc/gemm_f32.hip) -- and by
    torch's own float32 convolution, each against a float64 product: rms error relative to the output's largest magnitude.  The
    headline's dtype is f32 because these numbers say so, not because the operands are."""
    from openpifpaf_amd import fused
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn((2, 1024, 81, 81), device='cuda', generator=g).clamp_(min=0).contiguous(memory_format=torch.channels_last)
    w = torch.randn((256, 1024), device='cuda', generator=g) * (2.0 / 1024) ** 0.5
    b = torch.randn(256, device='cuda', generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), w.double().view(256, 1024, 1, 1), b.double()).clamp_(min=0)
    scale = float(ref.abs().max())

    def rms(out):
        d = out.double() - ref
        return float('%.3g' % (float((d * d).mean().sqrt()) / scale))
    out = {'shape': 'conv1x1 1024->256 @81x81 x2, relu', 'metric': 'rms error vs float64 / max|out|',
           'f32_mfma': rms(fused.conv1x1_bias_act(x, w, b, None, True)),
           'torch_f32_conv': rms(torch.relu(torch.nn.functional.conv2d(x, w.view(256, 1024, 1, 1), b)))}
    if fused.X3_TERMS:
        out['split_bf16_%dterms' % fused.X3_TERMS] = rms(fused.conv1x1_bias_act_x3(x, fused.split_weight(w), b, None, True, None, fused.X3_TERMS))
    return out


def kernel_source_hash():
    """Identifies the DECODE kernels a PMC traffic file was measured on (profiles/r*/pmc_traffic.json; the same function
    stamps it: tools/summarize_profiles.py)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'openpifpaf_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.hpp')) and name not in PRODUCER_SOURCES:
            h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(config_id, B, force_complete=False):
    """PMC HBM bytes of the WHOLE decode path per launch (rocprofv3 passes cannot run inside bench.py;
    tools/collect_profiles.sh writes them, stamped with the hash of the kernel sources they were measured on)."""
    pmc, where, have = None, None, kernel_source_hash()
    for rnd in PMC_ROUNDS:                                     # newest first; only a file measured on THESE kernels counts
        try:
            cand = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')))
        except (OSError, ValueError):
            continue
        if cand.get('kernel_source_hash') == have:
            pmc, where = cand, 'profiles/%s/pmc_traffic.json' % rnd
            break
    if pmc is None:
        return None, 'no pmc_traffic.json measured on these kernel sources (run tools/collect_profiles.sh)'
    key = 'config%d%s_batch%d' % (config_id, '_fc' if force_complete else '', B)
    entry = pmc.get('workloads', {}).get(key)
    same_as = ''
    if not entry and config_id == 3:
        # configs[2] differs from configs[1] in the network only: the decode sees the same field shapes, the same field
        # batches and the same launches, so the counters collected for config 2 are its counters
        key = 'config2%s_batch%d' % ('_fc' if force_complete else '', B)
        entry = pmc.get('workloads', {}).get(key)
        same_as = '; config 3 decodes the same field batches with the same launches as config 2'
    if not entry:
        return None, '%s holds no entry %s' % (where, key)
    return entry, '%s[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, x2 read ' \
                  'correction; sum over the decode kernels of one launch%s)' % (where, key, same_as)



# ----------------------------------------------------------------------------------------------- the printed line
LINE_LIMIT = 3072        # the driver keeps a stdout tail of a few KB: the final line must stay well inside it
DETAIL_FILE = 'bench_detail.json'


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_roofline(roof):
    if not roof:
        return None
    out = _pick(roof, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                       'algorithmic_bytes_per_launch'))
    path = roof.get('decode_path') or {}
    out['decode_path_frac'] = path.get('frac')
    out['decode_path_ms'] = path.get('ms_per_batch')
    if path.get('tie_pass_ms') == 0.0 and str(path.get('seed_tie_order', '')).startswith('libstdcxx'):
        out['kernel_includes'] = 'seed tie pass (r5: own launch, 0.078 ms)'
    return out


def compact_cpu(cpu):
    if not cpu:
        return None
    out = _pick(cpu, ('value', 'cores', 'kind', 'fresh_instance_value', 'all_cores_value', 'all_cores'))
    out['unit'] = 'images/s (decode only)'
    out['sample'] = str(cpu.get('sample_short', cpu.get('sample', '')))[:96]
    return out


def compact_parity(par):
    return _pick(par, ('images', 'poses', 'max_abs_delta', 'discrete_mismatches')) if par else None


def parity_ok(par):
    if not par:
        return None
    return bool(par.get('images', 0) > 0 and par.get('discrete_mismatches', 1) == 0
                and par.get('images_beyond_tolerance', 0) == 0 and par.get('max_abs_delta', 1.0) <= TOL)


def compact_leg(leg):
    """One entry of the line's `configs` digest: {value, ms_per_step, decode_ms, frac, parity_ok}."""
    if not isinstance(leg, dict):
        return None
    if 'error' in leg:
        return {'error': str(leg['error'])[:80]}
    roof = leg.get('roofline') or {}
    path = roof.get('decode_path') or {}
    value = leg.get('value', leg.get('decode_only_images_per_s'))
    if isinstance(value, dict):                               # the lanes sweep: its best entry
        value = max(value.values()) if value else None
    out = {'value': value,
           'ms_per_step': leg.get('ms_per_step', leg.get('ms_per_batch_wall', leg.get('eager_ms_per_image'))),
           'decode_ms': path.get('ms_per_batch', leg.get('decode_ms')),
           'frac': roof.get('frac', (leg.get('best') or {}).get('frac')),
           'parity_ok': parity_ok(leg.get('parity'))}
    if isinstance(leg.get('synchronous'), dict):              # the product path: pipelined (= value) and synchronous
        out['sync_value'] = leg['synchronous'].get('images_per_s')
    if 'vs_reference_cpu' in leg:                             # configs[0]: the reference's CPU-only predict beside it
        out['vs_ref_cpu'] = leg['vs_reference_cpu']
        out['ref_cpu_ms'] = (leg.get('reference_cpu') or {}).get('ms_per_image')
    if 'decode_vs_cpu_1thread' in leg and leg.get('cpu_baseline'):
        out['cpu_1thread'] = leg['cpu_baseline'].get('value')
    return out


def compact_line(detail):
    """The ONE line the driver parses, built from the full result (which goes to bench_detail.json): contract fields,
    roofline, cpu_baseline, parity and a five-number digest per extra leg -- no prose, always < LINE_LIMIT bytes."""
    line = _pick(detail, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                          'scaling', 'vs_baseline', 'dtype', 'data'))
    cfg = detail.get('config') or {}
    line['config'] = _pick(cfg, ('workload', 'backbone', 'backbone_dtype', 'conv3x3', 'conv1x1', 'decode_dtype', 'global_batch', 'batch_per_gpu',
                                 'parallelism', 'force_complete_pose'))
    line['roofline'] = compact_roofline(detail.get('roofline'))
    line['cpu_baseline'] = compact_cpu(detail.get('cpu_baseline'))
    line['parity'] = compact_parity(detail.get('parity'))
    if 'per_rank_ms_per_step' in detail:
        line['per_rank_ms_per_step'] = detail['per_rank_ms_per_step']
    if isinstance(detail.get('f32_check'), dict):
        line['f32_check'] = {k: v for k, v in detail['f32_check'].items() if k not in ('metric',)}
    bf = detail.get('bf16_backbone')
    if bf:
        line['bf16_backbone'] = _pick(bf, ('value', 'ms_per_step'))
    digest = {k: compact_leg(v) for k, v in (detail.get('configs') or {}).items()}
    if digest:
        line['configs'] = digest
    if detail.get('watchdog'):
        line['watchdog'] = str(detail['watchdog'])[:80]
    line['detail'] = DETAIL_FILE
    line['bench_seconds'] = detail.get('bench_seconds')
    text = json.dumps(line, separators=(',', ':'))
    if len(text) >= LINE_LIMIT:                               # never let an extra take the headline with it
        line.pop('configs', None)
        line['configs_dropped'] = True
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_detail(detail):
    """Everything the line leaves out: next to bench.py, and under gpurun_out/ so that it travels back from a GPU box."""
    text = json.dumps(detail, indent=1)
    written = []
    for path in (os.path.join(ROOT, DETAIL_FILE), os.path.join(ROOT, 'gpurun_out', DETAIL_FILE)):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as f:
                f.write(text)
            written.append(path)
        except OSError as e:                                  # read-only checkout: the line itself still prints
            print('bench: could not write %s: %r' % (path, e), file=sys.stderr)
    return written


_PARTIAL = {'line': None, 't0': time.perf_counter()}


def start_watchdog(soft_seconds):
    """A default run takes under three minutes.  Should a leg hang (a GPU wait that never returns, a forked worker stuck
    in a lock), the line must still get out: after `soft_seconds` the tracebacks of all threads go to stderr and, if the
    headline result exists, it is printed as the final line (marked) and the process exits; without it the exit is 4."""
    import faulthandler
    import threading

    def watch():
        time.sleep(soft_seconds)
        print('bench: watchdog after %.0f s -- a leg did not return; tracebacks follow' % soft_seconds, file=sys.stderr)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        line = _PARTIAL['line']
        if line is not None:
            line = dict(line)
            line['watchdog'] = 'an extra leg did not return within %d s; headline only' % soft_seconds
            line['bench_seconds'] = round(time.perf_counter() - _PARTIAL['t0'], 1)
            try:
                write_detail(line)
                print(compact_line(line), flush=True)
                os._exit(0)
            except BaseException as e:                 # (nothing printed: say so with the exit code)
                print('bench: watchdog could not print the line: %r' % (e,), file=sys.stderr)
        os._exit(4)
    t = threading.Thread(target=watch, daemon=True)
    t.start()
    return t


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (a bare run
    would silently measure one rank)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench: --gpus %d without WORLD_SIZE: re-launching as %s' % (n, ' '.join(cmd)), file=sys.stderr)
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


# ----------------------------------------------------------------------------------------------- CPU side
def reference_decoder(skeleton0, n_keypoints, fc_kw=None):
    """-> (kind, make_decode, torch) where make_decode(reuse) returns decode(cif, caf) on the host: the reference's
    own C++ decoder (oracle/_ref) when it is present, the restatement otherwise."""
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


def reset_reference():
    from oracle import reference
    if reference.available():
        reference.reset_statics()


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
    reused, n_reused = time_loop(make(True), cifs, cafs, seconds * 0.3, min(n, 16))
    fresh, n_fresh = time_loop(make(False), cifs, cafs, seconds * 0.3, min(n, 8))

    # all host cores: the reference's --decoder-workers mechanism is a fork pool
    # (reference decoder/decoder.py:33-47,130-131); N = os.cpu_count().  Time-bounded.
    cores = os.cpu_count() or 1
    multi = None
    try:
        import multiprocessing as mp
        ctx = mp.get_context('fork')
        budget = max(2.5, 0.4 * seconds)
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
        import queue as queue_mod
        results = []
        for _ in procs:                                # (a child forked from a threaded process can deadlock in an allocator lock:
            try:                                       # it must not take the whole run with it)
                results.append(q.get(timeout=max(1.0, t_end + 30.0 - time.perf_counter())))
            except queue_mod.Empty:
                break
        for pr in procs:
            pr.join(timeout=0.5)
            if pr.is_alive():
                pr.terminate()
        if len(results) < len(procs):
            print('cpu_baseline: %d of %d workers did not report' % (len(procs) - len(results), len(procs)), file=sys.stderr)
        multi = sum(k for k, _ in results) / (max(t for _, t in results) - t0) if results else None
    except Exception as e:   # pragma: no cover
        multi = None
        print('cpu_baseline: multi-process leg failed: %r' % (e,), file=sys.stderr)
    if fc_kw:
        reset_reference()
    return {
        'value': round(reused, 2), 'unit': 'images/s (decode only, 1 thread, decoder instance reused)', 'cores': 1,
        'kind': kind, 'fresh_instance_value': round(fresh, 2),
        'all_cores_value': round(multi, 2) if multi else None, 'all_cores': cores,
        'sample_short': '%d reused + %d fresh-instance decodes of the rank-0 field batch, then %d forked workers' % (
            n_reused, n_fresh, cores),
        'sample': '%d decodes of the rank-0 field batch with one reused decoder instance (what the reference\'s '
                  'Decoder does), %d with a fresh instance per image (the parity definition: revision drift), '
                  'then %d forked single-thread workers for a fixed time budget' % (n_reused, n_fresh, cores),
    }


def parity_stamp(decode_device, variants, skeleton0, n_keypoints, fc_kw=None, max_images=None):
    """Decode every field batch once more OUTSIDE the timed region and compare every image with the reference
    decoder (fresh instance per image = the parity definition).  -> dict for the JSON line."""
    from openpifpaf_amd import native
    kind, make, _ = reference_decoder(skeleton0, n_keypoints, fc_kw)
    decode_ref = make(False)
    images = mismatches = poses = differing = 0
    worst = 0.0
    t0 = time.perf_counter()
    for cifs_np, cafs_np, cif_d, caf_d in variants:
        out, ids, counts = decode_device(cif_d, caf_d)
        out, counts = out.cpu().numpy(), counts.cpu().numpy()
        native.check_counts(counts)
        n_img = len(counts) if max_images is None else min(len(counts), max_images)
        for b in range(n_img):
            want = decode_ref(cifs_np[b], cafs_np[b])
            want = np.asarray(want[0].numpy() if hasattr(want[0], 'numpy') else want[0])
            got = out[b, :int(counts[b]) & native.COUNT_ROWS_MASK]
            images += 1
            poses += len(want)
            if got.shape != want.shape or not np.array_equal(got[..., 0] > 0, want[..., 0] > 0):
                mismatches += 1
                differing += 1
                continue
            d = float(np.abs(got.astype(np.float64) - want).max()) if got.size else 0.0
            worst = max(worst, d)
            differing += d > TOL
    if fc_kw:
        reset_reference()
    return {'images': images, 'poses': poses, 'max_abs_delta': worst, 'discrete_mismatches': mismatches,
            'images_beyond_tolerance': int(differing), 'tolerance': TOL, 'against': kind,
            'field_batches': len(variants), 'seconds': round(time.perf_counter() - t0, 2)}


# ----------------------------------------------------------------------------------------------- the workload
class Workload:
    """Field batches, decoder and host buffers of one BASELINE configuration on this rank."""

    def __init__(self, config_id, B, rank, device, long_edge, n_variants, backbone=None, max_annotations=None, full_pool=False):
        from openpifpaf_amd import constants, headmeta, native, synth
        cfg = CONFIGS[config_id]
        self.config_id, self.cfg, self.B, self.device = config_id, cfg, B, device
        self.backbone = backbone or cfg['backbone']
        if cfg['wholebody']:
            wb = constants.wholebody()
            self.cif_meta, self.caf_meta = headmeta.wholebody_metas()
            skeleton1, pose, self.people = wb['skeleton'], wb['standing_pose'], (1, 3, 6, 10)
        else:
            self.cif_meta, self.caf_meta = headmeta.cocokp_metas()
            skeleton1, pose, self.people = constants.COCO_PERSON_SKELETON, None, synth.PEOPLE_CYCLE
        self.skeleton0 = np.asarray(skeleton1, dtype=np.int64) - 1
        self.K, self.A = self.cif_meta.n_fields, self.caf_meta.n_fields
        self.stride = self.cif_meta.stride
        self.long_edge = long_edge
        self.fh = (long_edge - 1) // 16 * 2 + 1            # 641 -> 41 -> 82 -> 81
        self.variants = []
        for v in range(n_variants):
            cifs, cafs = synth.synth_batch(B, seed0=v * VARIANT_SEED + rank * B, height=self.fh, width=self.fh,
                                           people=self.people, pose=pose,
                                           skeleton=skeleton1 if cfg['wholebody'] else None)
            self.variants.append((cifs, cafs, torch.from_numpy(cifs).to(device), torch.from_numpy(cafs).to(device)))
        self._quantised = None
        kw = {} if max_annotations is None else {'max_annotations': max_annotations}
        if full_pool:
            kw['cifhr_pool_tiles'] = 'full'
        self.dec = native.CifCaf(self.K, torch.from_numpy(self.skeleton0), **kw)
        self.host_out = torch.empty((B, self.dec.max_annotations, self.K, 4), dtype=torch.float32).pin_memory()
        self.host_counts = torch.empty((B,), dtype=torch.int32).pin_memory()

    def quantised(self):
        """The field batches rounded to bfloat16 (nearest even) and held as float32: what a bf16 head emits."""
        if self._quantised is None:
            self._quantised = []
            for cifs, cafs, cif_d, caf_d in self.variants:
                cq, fq = cif_d.to(torch.bfloat16).to(torch.float32), caf_d.to(torch.bfloat16).to(torch.float32)
                self._quantised.append((cq.cpu().numpy(), fq.cpu().numpy(), cq, fq))
        return self._quantised

    def workload_text(self):
        return ('BASELINE %s: %s %dx%d, batch %d per GPU, CIF/CAF fields [%d,%d,5,%d,%d]+[%d,%d,8,%d,%d]'
                % (self.cfg['name'], self.backbone, self.long_edge, self.long_edge, self.B, self.B, self.K, self.fh,
                   self.fh, self.B, self.A, self.fh, self.fh))


def _synth_part(seed0, n, fh):
    from openpifpaf_amd import synth
    return synth.synth_batch(n, seed0=seed0, height=fh, width=fh)


def Workload_view(wl, B, variants):
    """The same decoder and shapes with another batch size / field batches (kernel_profile only reads these)."""
    v = Workload.__new__(Workload)
    v.__dict__.update(wl.__dict__)
    v.B, v.variants = B, variants
    return v


def build_model(wl, dtype_name):
    from openpifpaf_amd import network
    model = network.factory(wl.backbone, [wl.cif_meta, wl.caf_meta]).to(wl.device)
    network.optimize_for_inference_(model)
    model = model.to(memory_format=torch.channels_last)
    if dtype_name != 'fp32':
        model = model.to(TORCH_DTYPE[dtype_name])
    return model


def native_tie_order():
    from openpifpaf_amd import native
    return native.get_seed_tie_order()


def kernel_profile(wl, variants, params, steps):
    """Per-kernel HIP-event times of the decode (events recorded by the library on the stream the kernels are launched
    on), averaged over `steps` launches that alternate the field batches."""
    from openpifpaf_amd import _lib, native
    per_kernel = {}
    for i in range(len(variants)):                             # untimed: the first launches after the network leg
        _, _, cif_d, caf_d = variants[i % len(variants)]       # (event pool, clocks) are not the steady state
        wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride, params=params)
    for i in range(max(2, steps)):
        _, _, cif_d, caf_d = variants[i % len(variants)]
        _lib.profile_begin(native._stream())
        wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride, params=params)
        for name, ms in _lib.profile_end():
            per_kernel.setdefault(name, []).append(ms)
    return {k: float(np.mean(v)) for k, v in per_kernel.items()}


def tiles_written_per_call(wl, variants, params):
    """32x64 tiles of the CifHr map one call writes (mean over the alternating field batches): the tiles this call's CIF
    cells reach -- the map is a pool of tiles, nothing is cleared and nothing carries over -- counted from the touched-tile
    bitmaps (second half of the workspace's tile state)."""
    rows = (wl.fh - 1) * wl.stride + 1
    words = ((((rows + 63) // 64) * ((rows + 31) // 32)) + 31) // 32
    n = wl.B * wl.K * words
    counts = []
    for _, _, cif_d, caf_d in variants:
        wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride, params=params)
        bm = wl.dec.workspace_view('tile_bitmaps', torch.int32).cpu().numpy().view(np.uint32)
        counts.append(int(np.unpackbits(bm[n:2 * n].copy().view(np.uint8)).sum()))
    return int(round(float(np.mean(counts))))


def decode_roofline(wl, variants, params, steps, force_complete=False):
    avg_ms = kernel_profile(wl, variants, params, steps)
    alg = algorithmic_bytes(wl.B, wl.K, wl.A, wl.fh, wl.fh, wl.stride, wl.dec.max_annotations)
    alg['cifhr_tile_kernel'] = tiles_written_per_call(wl, variants, params) * 32 * 64 * 4
    decode_ms = sum(avg_ms.values())
    dominant = max(avg_ms, key=avg_ms.get)
    dom_ms = avg_ms[dominant]
    achieved = alg['decode_path'] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0        # SURVEY 8d bytes per launch
    own = alg.get(dominant, 0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    entry, note = pmc_traffic(wl.config_id, wl.B, force_complete)
    roofline = {
        'bound': 'hbm', 'kernel': dominant, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBPS,
        'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 5),
        'traffic': entry['decode_path_hbm_bytes'] if entry else None, 'traffic_source': note,
        'traffic_definition': 'PMC HBM bytes of the WHOLE decode path per launch (comparable with '
                              'algorithmic_bytes_per_launch); per kernel under traffic_per_kernel',
        'traffic_per_kernel': entry.get('kernels') if entry else None,
        'avg_launch_ms': round(dom_ms, 4), 'algorithmic_bytes_per_launch': alg['decode_path'],
        'algorithmic_bytes_definition': 'SURVEY 8d: CIF + CAF + annotations of one image x images per launch',
        'own_bytes': {'bytes_per_launch': alg.get(dominant, 0), 'GBps': round(own, 2),
                      'frac': round(own / HBM_PEAK_GBPS, 6),
                      'note': 'the kernel\'s own compulsory bytes (the association kernels only write the '
                              'annotations: a latency-bound dependency chain, not a bandwidth problem)'},
        'kernels': {k: {'ms': round(v, 4),
                        'GBps': round(alg.get(k, 0) / (v * 1e-3) / 1e9, 1) if v > 0 else None}
                    for k, v in avg_ms.items()},
        'decode_path': {'ms_per_batch': round(decode_ms, 4),
                        'images_per_s': round(wl.B / (decode_ms * 1e-3), 1),
                        'GBps': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9, 2),
                        'frac': round(alg['decode_path'] / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                        'seed_tie_order': native_tie_order(),
                        'tie_pass_ms': round(avg_ms.get('cifseeds_tie_kernel', 0.0), 4),
                        'tie_pass_note': 'the pass that puts seeds of EQUAL score into the order the reference\'s unstable std::sort '
                                         'leaves them in runs INSIDE the association kernel since round 6 (tie_pass_ms 0: no '
                                         'launch of its own; rounds 3-5: cifseeds_tie_kernel, 0.078 ms per batch of 32) -- the '
                                         'dominant kernel\'s time includes it for the images that hold such seeds (the synthetic '
                                         'blobs are symmetric: about one image in eight); opa_cifcaf_set_tie_placement(dec, 0) '
                                         'brings the launch back'},
        'field_batches_alternating': len(variants),
    }
    return roofline


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        if torch.cuda.device_count() < args.gpus and not args.share_device:
            sys.exit('bench.py: --gpus %d but %d visible' % (args.gpus, torch.cuda.device_count()))
        spawn_ranks(args.gpus)
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
    if args.gpus != world:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE %d' % (args.gpus, world))

    from openpifpaf_amd import _lib, distributed, native

    t_program = time.perf_counter()
    if rank == 0:
        start_watchdog(args.watchdog_seconds)
    config_id = args.config or 2
    extras = (args.config is None and world == 1 and not args.no_extras and not args.decode_only
              and args.fields == 'synthetic' and not args.force_complete and args.backbone is None and args.batch is None)
    n_variants = 1 if (args.single_batch or args.graph) else 2
    wl = Workload(config_id, args.batch or CONFIGS[config_id]['batch'], rank, device, args.long_edge, n_variants,
                  backbone=args.backbone, full_pool=args.fields == 'network')
    B, K, A, stride = wl.B, wl.K, wl.A, wl.stride
    dec = wl.dec
    dec_params = _lib.default_params(**FC_KW) if args.force_complete else None
    main_stream = torch.cuda.current_stream()
    # high priority: the few decode workgroups slip in between the backbone's waves instead of queueing behind them
    dec_stream = main_stream if args.no_overlap else torch.cuda.Stream(priority=-1)

    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    images32 = torch.randn((B, 3, args.long_edge, args.long_edge), generator=g).to(device)
    images32 = images32.contiguous(memory_format=torch.channels_last)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    def run_leg(wl, images32, dtype_name, steps, warmup, *, params=None, decode_only=False, fields='synthetic',
                quantised=False, n_streams=1, graph=False, dump=None, tie_inside=None):
        """W warm-up steps, then EXACTLY `steps` timed steps between barrier + synchronize.  -> dict(elapsed = max over
        ranks, per-rank times, model, images, annotations of the last step)."""
        model = None if decode_only else build_model(wl, dtype_name)
        images = images32 if dtype_name == 'fp32' else images32.to(TORCH_DTYPE[dtype_name])
        variants = wl.quantised() if quantised else wl.variants
        lanes = native.DecodeLanes(wl.K, torch.from_numpy(wl.skeleton0), lanes=n_streams) if n_streams > 1 else None
        if lanes is not None and tie_inside is not None:        # (DecodeLanes puts the tie pass inside the association kernel for lanes > 1)
            for d_ in lanes.decoders:
                d_.set_tie_placement(tie_inside)
        graphs = None
        if graph:                                          # one captured decode per (decoder, stream), one field batch
            decs = [wl.dec] + [native.CifCaf(wl.K, torch.from_numpy(wl.skeleton0)) for _ in range(n_streams - 1)]
            graphs = []
            for d in decs:
                st = torch.cuda.Stream(priority=-1)
                gr, outs = d.capture(variants[0][2], wl.stride, variants[0][3], wl.stride, params=params, stream=st)
                graphs.append((gr, st, outs))
        step_no = [0]
        shapes_checked = [False]
        gathered = [None]
        tickets = []

        def step():
            heads = None
            if model is not None:
                with torch.no_grad():
                    heads = model(images)
                if not shapes_checked[0]:
                    assert tuple(heads[0].shape) == tuple(variants[0][2].shape), (heads[0].shape, variants[0][2].shape)
                    assert tuple(heads[1].shape) == tuple(variants[0][3].shape), (heads[1].shape, variants[0][3].shape)
                    shapes_checked[0] = True
            _, _, cif_d, caf_d = variants[step_no[0] % len(variants)]      # the field batches alternate
            step_no[0] += 1
            if graphs is not None:
                gr, st, (out, ids, counts) = graphs[step_no[0] % len(graphs)]
                with torch.cuda.stream(st):
                    gr.replay()
                    wl.host_out.copy_(out, non_blocking=True)
                    wl.host_counts.copy_(counts, non_blocking=True)
                return
            if lanes is not None:                          # decode-only: several batches in flight
                t = lanes.submit(cif_d, wl.stride, caf_d, wl.stride, params=params)
                tickets.append(t)
                if len(tickets) > n_streams:               # the host copy of the oldest, on its lane's stream order
                    out, ids, counts = tickets.pop(0).result()
                    wl.host_out.copy_(out, non_blocking=True)
                    wl.host_counts.copy_(counts, non_blocking=True)
                return
            ev = torch.cuda.Event()
            ev.record(main_stream)
            with torch.cuda.stream(dec_stream):
                dec_stream.wait_event(ev)                  # decode of batch i follows its backbone
                if fields == 'network' and model is not None:
                    out, ids, counts = wl.dec.call_batch(heads[0], wl.stride, heads[1], wl.stride, params=params)
                else:
                    out, ids, counts = wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride, params=params)
                if world > 1:                              # final annotations only, ONE collective (RCCL over xGMI)
                    if args.dist_backend == 'nccl':
                        gathered[0] = distributed.gather_annotations(out, ids, counts)
                    else:
                        gathered[0] = distributed.gather_annotations(out.cpu(), ids.cpu(), counts.cpu())
                wl.host_out.copy_(out, non_blocking=True)
                wl.host_counts.copy_(counts, non_blocking=True)

        t_setup = time.perf_counter()
        for _ in range(warmup):
            step()
        sync_all()
        # (round 6: no broadcast of 1x1-convolution kernel choices any more -- the package ships the table for these shapes,
        # fused.load_pinned, and a multi-rank job takes the same default for a shape the table does not know)
        if rank == 0:
            print('bench: config %d %s leg warm-up (incl. MIOpen find) %.1f s' % (wl.config_id, dtype_name,
                                                                                  time.perf_counter() - t_setup), file=sys.stderr)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        for t in tickets:
            t.result()
        sync_all()
        elapsed = time.perf_counter() - t0
        per_rank = [elapsed]
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.dist_backend == 'nccl' else 'cpu')
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank = [float(x.item()) for x in every]
            elapsed = max(per_rank)
        native.check_counts(wl.host_counts)
        if dump and rank == 0:
            a_, i_, c_ = gathered[0] if gathered[0] is not None else (wl.host_out, None, wl.host_counts)
            np.savez(dump, annotations=a_.cpu().numpy(), counts=c_.cpu().numpy(),
                     variant=(step_no[0] - 1) % len(variants), variant_seed=VARIANT_SEED)
        return dict(elapsed=elapsed, per_rank=per_rank, model=model, images=images,
                    n_ann=int((wl.host_counts & native.COUNT_ROWS_MASK).sum()), variants=variants)

    def network_only(model, images, wl, reps=5):
        """Network alone (backbone + heads) per batch, the host copy of its field tensors the reference makes before
        decoding (decoder/decoder.py:96-100), and the heads of the last forward."""
        from openpifpaf_amd import winograd
        winograd.reset_flop_counter()
        with torch.no_grad():
            heads = model(images)
        network_only.winograd_gflop = winograd.direct_flops() / 1e9     # direct-form flops of ONE forward that F(2x2, 3x3) ran
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

    def reference_flow_run(model, images, wl, steps, fc_kw=None):
        """The reference's data flow, RUN (not composed): network on the GPU -> .cpu() of its head fields -> the
        reference C++ decoder on one host thread, one instance reused, image after image (the decode works on the
        synthetic fields of the same shapes: a random-init head's own output decodes ~30x slower)."""
        kind, make, _ = reference_decoder(wl.skeleton0, wl.K, fc_kw)
        decode = make(True)
        cifs, cafs = wl.variants[0][0], wl.variants[0][1]
        n = len(images)
        with torch.no_grad():
            [h.cpu() for h in model(images)]
        t0 = time.perf_counter()
        for s in range(steps):
            with torch.no_grad():
                host = [h.cpu() for h in model(images)]                 # blocks until the network is done
            assert host[0].shape[0] == n
            for b in range(n):
                decode(cifs[b % len(cifs)], cafs[b % len(cafs)])
        dt = (time.perf_counter() - t0) / steps
        if fc_kw:
            reset_reference()
        return {'images_per_s_1thread': round(n / dt, 2), 'ms_per_batch': round(dt * 1e3, 1), 'steps': steps,
                'decoder': kind, 'measured': 'run end to end in this process'}

    def leg_summary(wl, leg, steps):
        return dict(value=world * wl.B * steps / leg['elapsed'], ms_per_step=leg['elapsed'] / steps * 1e3)

    def full_config(wl, images32, steps, warmup, *, primary, bf16_leg, cpu_seconds, profile_steps, params, dump=None,
                    decode_only=False, fields='synthetic', n_streams=1, graph=False, reference_steps=3, fc=False):
        """All legs of one configuration on this rank -> result dict (rank 0) or None."""
        legs, nn, ref_run, wino = {}, {}, {}, {}
        leg = run_leg(wl, images32, primary, steps, warmup, params=params, decode_only=decode_only, fields=fields,
                      n_streams=n_streams, graph=graph, dump=dump)
        legs[primary] = leg
        single = rank == 0 and world == 1
        if leg['model'] is not None and single:
            nn[primary] = network_only(leg['model'], leg['images'], wl)
            wino[primary] = network_only.winograd_gflop
            if not args.no_cpu_baseline:
                ref_run[primary] = reference_flow_run(leg['model'], leg['images'], wl, reference_steps, FC_KW if fc else None)
        leg['model'] = None
        torch.cuda.empty_cache()
        if bf16_leg and not decode_only and primary != 'bf16':
            leg2 = run_leg(wl, images32, 'bf16', steps, warmup, params=params, quantised=True)
            legs['bf16'] = leg2
            if single:
                nn['bf16'] = network_only(leg2['model'], leg2['images'], wl)
                wino['bf16'] = network_only.winograd_gflop
                if not args.no_cpu_baseline:
                    ref_run['bf16'] = reference_flow_run(leg2['model'], leg2['images'], wl, reference_steps, FC_KW if fc else None)
            leg2['model'] = None
            torch.cuda.empty_cache()
        if rank != 0:
            return None
        fc_kw = FC_KW if fc else None
        with torch.cuda.stream(dec_stream):
            roofline = decode_roofline(wl, wl.variants, params, profile_steps, force_complete=fc)
        if not decode_only and not wl.cfg['wholebody'] and wl.backbone == 'resnet50' and nn:
            gflop = 274.0 * wl.B                          # 137 GMAC per 641x641 image
            # [r6] the stride-1 3x3 convolutions of the float32 trunk run as Winograd F(2x2, 3x3): of their direct-form flops
            # (counted by openpifpaf_amd.winograd per forward) only 1 / 2.25 execute.  TFLOPs / frac_of_dense_peak are what the
            # MFMA units EXECUTE; TFLOPs_direct_equivalent is the direct-form figure a convolution benchmark would quote.
            executed = {d: gflop - wino.get(d, 0.0) * (1.0 - 1.0 / 2.25) for d in nn}
            roofline['backbone_mfma'] = {
                d: {'ms_per_batch': round(nn[d][0], 2), 'TFLOPs': round(executed[d] / nn[d][0], 1),
                    'frac_of_dense_peak': round(executed[d] / nn[d][0] / MFMA_PEAK_TFLOPS[d], 3),
                    'TFLOPs_direct_equivalent': round(gflop / nn[d][0], 1),
                    'winograd_direct_gflop_per_batch': round(wino.get(d, 0.0), 1)} for d in nn}
        cpu = ref_pipe = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(wl.variants[0][0], wl.variants[0][1], wl.skeleton0, wl.K, cpu_seconds, fc_kw)
            if nn:
                ref_pipe = {}
                for d, (nn_ms, d2h_ms, field_bytes) in nn.items():
                    allc = wl.B / ((nn_ms + d2h_ms) * 1e-3 + wl.B / cpu['all_cores_value']) if cpu.get('all_cores_value') else None
                    ref_pipe[d] = dict(ref_run.get(d, {}))
                    ref_pipe[d].update({
                        'network_ms_per_batch': round(nn_ms, 2), 'fields_to_host_ms_per_batch': round(d2h_ms, 2),
                        'field_bytes_per_batch': field_bytes,
                        'cpu_decode_ms_per_batch_1thread': round(wl.B / cpu['value'] * 1e3, 1),
                        'images_per_s_all_cores_composed': round(allc, 1) if allc else None, 'cores': cpu['all_cores'],
                    })
        parity = None
        if world == 1 and not args.no_parity:
            with torch.cuda.stream(dec_stream):
                parity = parity_stamp(lambda c, f: wl.dec.call_batch(c, wl.stride, f, wl.stride, params=params),
                                      wl.variants, wl.skeleton0, wl.K, fc_kw)
                if 'bf16' in legs:
                    q = parity_stamp(lambda c, f: wl.dec.call_batch(c, wl.stride, f, wl.stride, params=params),
                                     wl.quantised(), wl.skeleton0, wl.K, fc_kw)
                    q['seed_tie_order'] = native_tie_order()
                    q['note'] = ('decode inputs of the bf16 leg: the fields rounded to bfloat16.  8-bit mantissas make seed '
                                 'scores tie, and the reference orders tied seeds by whatever its unstable std::sort leaves '
                                 '(cif_seeds.cpp:94).  seed_tie_order "libstdcxx": the HIP path reproduces that order '
                                 '(cifseeds_tie_kernel); "index": it orders them by cell index, and '
                                 'images_beyond_tolerance / images is the tie mismatch rate (round 2: 5 of 64)')
                    parity['bf16_fields'] = q
        s = leg_summary(wl, legs[primary], steps)
        value = s['value']
        base = ref_pipe[primary].get('images_per_s_1thread') if ref_pipe and primary in ref_pipe else None
        result = {
            'value': round(value, 2), 'unit': 'images/s', 'ms_per_step': round(s['ms_per_step'], 3),
            'vs_baseline': round(value / base, 2) if base else None,
            'config': {
                'workload': wl.workload_text(),
                'backbone': 'none (decode only)' if decode_only else wl.backbone,
                'backbone_dtype': primary, 'decode_dtype': 'f32 (+f64 where the reference uses double)',
                'conv3x3': ('winograd F(2x2,3x3) f32 for the stride-1 ones' if wino.get(primary) else 'direct (MIOpen)'),
                'conv1x1': _conv1x1_text(primary),
                'global_batch': world * wl.B, 'batch_per_gpu': wl.B,
                'fields': ('COCO-shaped synthetic fields injected after the heads (people per image cycle %s); %d different '
                           'field batches alternate step by step' % (list(wl.people), len(wl.variants)))
                          if fields == 'synthetic' else
                          'the random-init network\'s own head outputs (all-active adversarial case)',
                'parallelism': 'images sharded one batch per GPU (dp%d); one RCCL all_gather of the packed annotations' % world
                               if world > 1 else 'single GPU',
                'decode_overlapped_on_second_stream': not args.no_overlap,
                'decode_streams': n_streams, 'hip_graph': bool(graph),
                'force_complete_pose': bool(fc),
                'annotations_per_batch': legs[primary]['n_ann'],
            },
            'roofline': roofline, 'cpu_baseline': cpu, 'reference_pipeline': ref_pipe, 'parity': parity,
        }
        if world > 1:
            pr = legs[primary]['per_rank']
            result['per_rank_ms_per_step'] = {'min': round(min(pr) / steps * 1e3, 3), 'max': round(max(pr) / steps * 1e3, 3)}
        if 'bf16' in legs and primary != 'bf16':
            s16 = leg_summary(wl, legs['bf16'], steps)
            b16 = ref_pipe['bf16'].get('images_per_s_1thread') if ref_pipe and 'bf16' in ref_pipe else None
            result['bf16_backbone'] = {
                'value': round(s16['value'], 2), 'ms_per_step': round(s16['ms_per_step'], 3),
                'vs_baseline_same_precision': round(s16['value'] / b16, 2) if b16 else None,
                'vs_reference_fp32_pipeline': round(s16['value'] / base, 2) if base else None,
                'note': 'same step with the network in bfloat16 and the decode inputs rounded to bfloat16 (reduced precision '
                        'relative to the reference; not the headline)',
            }
        if cpu is not None:
            result['decode_vs_cpu_1thread'] = round(roofline['decode_path']['images_per_s'] / cpu['value'], 1)
        return result

    # ------------------------------------------------------------------------------------- the headline
    primary = args.backbone_dtype
    result = full_config(wl, images32, args.steps, args.warmup, primary=primary,
                         bf16_leg=not args.no_bf16_leg, cpu_seconds=args.cpu_seconds,
                         profile_steps=args.profile_steps, params=dec_params, dump=args.dump_annotations,
                         decode_only=args.decode_only, fields=args.fields,
                         n_streams=max(1, args.decode_streams) if args.decode_only else 1, graph=args.graph and args.decode_only,
                         fc=args.force_complete)
    line = None
    if rank == 0:
        line = {
            'metric': ('images/sec end-to-end (backbone+CifCaf decode), %s 641px' % wl.backbone if not args.decode_only else
                       'images/sec DECODE ONLY (diagnostic: backbone skipped, not the headline metric)'),
            'value': result['value'], 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': result['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': result['vs_baseline'],
            'vs_baseline_definition': 'value / the reference data flow RUN in this process at the same backbone '
                                      'precision: network on the MI355X -> .cpu() of the fields -> reference C++ '
                                      'decoder, 1 host thread (reference_pipeline)',
            'dtype': 'f32', 'data': 'synthetic',
        }
        for k in ('config', 'roofline', 'cpu_baseline', 'reference_pipeline', 'parity', 'bf16_backbone',
                  'decode_vs_cpu_1thread', 'per_rank_ms_per_step'):
            if k in result:
                line[k] = result[k]
        rp = result.get('reference_pipeline') or {}
        if primary in rp and 'ms_per_batch' in rp[primary]:
            nn_ms, ref_ms = rp[primary]['network_ms_per_batch'], rp[primary]['ms_per_batch']
            line['target_15x'] = {
                'reachable_at_reference_precision': False,
                'bound': round(ref_ms / nn_ms, 2),
                'why': 'north_star asks >= 15x end to end vs the reference CPU decode path.  Both flows run the SAME %s '
                       'network on the MI355X (%.1f ms per batch); the reference flow then spends %.1f ms on the host copy '
                       'and the CPU decode.  Even a decode that costs nothing gives %.1f / %.1f = %.2fx; every millisecond '
                       'taken off the network is taken off both flows.  15x needs a faster network than the reference\'s '
                       'float32 one: see bf16_backbone (reduced precision, reported beside the headline, not as it).'
                       % (primary, nn_ms, ref_ms - nn_ms, ref_ms, nn_ms, ref_ms / nn_ms)}

        if primary == 'fp32' and not args.decode_only:
            try:
                line['f32_check'] = f32_numerics_check()
            except Exception as e:            # noqa: BLE001  (a check, not the measurement)
                line['f32_check'] = {'error': repr(e)[:80]}

    if rank == 0:
        _PARTIAL['line'] = line
    # ------------------------------------------------------------------------------------- the other configurations
    if extras:
        others = {}
        del images32
        torch.cuda.empty_cache()

        def guarded(name, fn):
            t0 = time.perf_counter()
            try:
                others[name] = fn()
            except Exception as e:            # an extra leg must not take the headline line with it
                import traceback
                traceback.print_exc()
                others[name] = {'error': repr(e)}
            if isinstance(others[name], dict):
                others[name]['leg_seconds'] = round(time.perf_counter() - t0, 1)


        # FIRST of the extra legs: behind the lane sweeps, which leave dozens of HIP streams behind, the same loop measured
        # 8 % slower (round 4: the driver's record carried that figure)
        # the product path: Predictor -> Decoder.batch_async over decode lanes (what a user of openpifpaf.predict gets), fed
        # uint8 frames that are preprocessed on the device; the network runs for real, the decode sees the headline's
        # synthetic fields (a random-init head's own output is the all-active case above)
        def predictor_leg():
            from openpifpaf_amd import Predictor, predictor as predictor_mod
            net = build_model(wl, 'fp32')

            class Injected(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.net, self.head_metas, self.n = net, [wl.cif_meta, wl.caf_meta], 0

                def forward(self, x):
                    self.net(x)
                    v = wl.variants[self.n % len(wl.variants)]
                    self.n += 1
                    return (v[2], v[3])
            saved = (Predictor.batch_size, Predictor.long_edge, Predictor.device_preprocess, Predictor.device)
            Predictor.batch_size, Predictor.long_edge, Predictor.device_preprocess, Predictor.device = wl.B, args.long_edge, True, device
            try:
                pred = Predictor(model=Injected())
                rng = np.random.default_rng(3)
                frames = [rng.integers(0, 255, (args.long_edge, args.long_edge, 3), dtype=np.uint8) for _ in range(wl.B)]
                n_batches = 24                             # (a short run is dominated by filling and draining the pipeline)
                out = {}
                for mode, pipelined in (('pipelined', True), ('synchronous', False)):
                    pred.pipelined = pipelined
                    list(pred.numpy_images(frames * 2))                     # warm-up (lanes, pinned buffers)
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    n_ann = sum(len(p) for p, _, _ in pred.numpy_images(frames * n_batches))
                    dt = time.perf_counter() - t0
                    out[mode] = {'images_per_s': round(wl.B * n_batches / dt, 1), 'ms_per_batch': round(dt / n_batches * 1e3, 2),
                                 'annotations': n_ann}
                # where a batch's time goes when nothing overlaps
                t0 = time.perf_counter()
                batch, metas = pred._preprocess(frames)
                torch.cuda.synchronize(device)
                out['preprocess_ms_per_batch'] = round((time.perf_counter() - t0) * 1e3, 2)
                t0 = time.perf_counter()
                res = pred.tensor_batch(batch, metas)
                out['tensor_batch_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
                out['value'] = out['pipelined']['images_per_s']
                out['ms_per_step'] = out['pipelined']['ms_per_batch']
                # The same pipelined loop in a process of its own (tools/gpu/predictor_probe.py: what a user's process looks
                # like -- no other model, no other legs' streams and allocations): inside this process the loop has measured
                # 8 % slower in two rounds running, wherever in the run it sits.  That figure is the leg's value; the
                # in-process ones stay beside it.
                try:
                    import re
                    import subprocess
                    probe = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gpu', 'predictor_probe.py'), '--batches', '24'],
                                           capture_output=True, text=True, timeout=300)
                    m = re.search(r'wall ([0-9.]+) ms per batch \(([0-9.]+) images/s\)', probe.stdout)
                    if m:
                        out['fresh_process'] = {'ms_per_batch': float(m.group(1)), 'images_per_s': float(m.group(2)),
                                                'log': [ln for ln in probe.stdout.splitlines() if ln.strip()][-3:]}
                        out['value'], out['ms_per_step'] = float(m.group(2)), float(m.group(1))
                    else:
                        out['fresh_process'] = {'error': (probe.stderr or probe.stdout)[-300:]}
                except Exception as e:                     # noqa: BLE001 -- the in-process figures stand
                    out['fresh_process'] = {'error': repr(e)}
                out['lanes'] = pred.processor.pipeline_depth
                out['what'] = ('Predictor.numpy_images over %d batches of %d uint8 %dx%d frames: device-side preprocessing, float32 %s, '
                               'HIP decode of the synthetic fields on decode lanes, inverse transform on the device, annotations '
                               'through pinned memory, Annotation objects built on the host' % (
                                   n_batches, wl.B, args.long_edge, args.long_edge, wl.backbone))
                return out
            finally:
                Predictor.batch_size, Predictor.long_edge, Predictor.device_preprocess, Predictor.device = saved
        guarded('predictor', predictor_leg)

        # the reference benchmark CLI's decoder setting on the headline's fields (decode only: the network is the same)
        def fc_leg():
            fc_params = _lib.default_params(**FC_KW)
            leg = run_leg(wl, None, 'fp32', 10, 2, params=fc_params, decode_only=True)
            with torch.cuda.stream(dec_stream):
                roof = decode_roofline(wl, wl.variants, fc_params, args.profile_steps, force_complete=True)
                par = None if args.no_parity else parity_stamp(
                    lambda c, f: wl.dec.call_batch(c, wl.stride, f, wl.stride, params=fc_params), wl.variants,
                    wl.skeleton0, wl.K, FC_KW)
            cpu = None if args.no_cpu_baseline else cpu_baseline(wl.variants[0][0], wl.variants[0][1], wl.skeleton0,
                                                                 wl.K, 6.0, FC_KW)
            return {'setting': 'reference benchmark.py:77-79: --force-complete-pose and zero keypoint / instance thresholds '
                               '(decoder/cifcaf.py:180-185); decode only, config 2 fields',
                    'decode_only_images_per_s': round(wl.B * 10 / leg['elapsed'], 1),
                    'ms_per_batch_wall': round(leg['elapsed'] / 10 * 1e3, 3), 'annotations_per_batch': leg['n_ann'],
                    'roofline': roof, 'cpu_baseline': cpu, 'parity': par,
                    'decode_vs_cpu_1thread': round(roof['decode_path']['images_per_s'] / cpu['value'], 1) if cpu else None}
        guarded('force_complete', fc_leg)

        # several batches in flight (stages 1-5 of batch i+1 beside the association of batch i: an association launch
        # keeps 32 of the 256 compute units busy)
        def lanes_leg():
            alg = algorithmic_bytes(wl.B, wl.K, wl.A, wl.fh, wl.fh, wl.stride, wl.dec.max_annotations)['decode_path']
            rate, gbps = {}, {}
            for inside in (None, False):          # default (DecodeLanes: the tie pass inside the association kernel) / a launch of its own
                for n, key in ((1, 'one_in_flight'), (2, 'two_in_flight'), (4, 'four_in_flight'), (8, 'eight_in_flight'),
                               (12, 'twelve_in_flight')):
                    if inside is False and n in (1, 4):
                        continue
                    key = key if inside is None else key + '_ties_launch'
                    steps = 96                             # (every lane's first calls allocate its workspace: warm-up per lane)
                    leg = run_leg(wl, None, 'fp32', steps, 4 * n + 4, decode_only=True, n_streams=n, tie_inside=inside)
                    rate[key] = round(wl.B * steps / leg['elapsed'], 1)
                    gbps[key] = round(alg * steps / leg['elapsed'] / 1e9, 1)
            best = max(gbps, key=gbps.get)
            return {'decode_only_images_per_s': rate, 'GBps': gbps,
                    'best': {'mode': best, 'GBps': gbps[best], 'frac': round(gbps[best] / HBM_PEAK_GBPS, 5)},
                    'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default'),
                    'what': 'decode only, config 2 fields, alternating batches, annotations copied to the host; '
                            'native.DecodeLanes(lanes=n): n decoders / workspaces / streams; GBps = SURVEY 8d '
                            'algorithmic bytes of a batch x batches per second (wall clock, whole decode path); '
                            'two or more lanes run the tie pass inside the association kernel (opa_cifcaf_set_tie_placement, the '
                            'DecodeLanes default); *_ties_launch: the same lanes with the pass as a launch of its own'}
        guarded('decode_two_in_flight', lanes_leg)

        # ONE call_batch of 256 images (eight of the headline's field batches: BASELINE configs[4]'s global batch on one GPU): the
        # launch that fills the chip -- the association kernel's 256 workgroups on 256 compute units -- per kernel, its roofline,
        # every image against the reference; the same with two and four decode lanes, and 512 images in one call (the
        # association kernel's image queue: more images than compute units, most seeds first)
        def b256_leg():
            from openpifpaf_amd import synth
            Bb = 256
            import multiprocessing as mp
            ctx = mp.get_context('fork')
            seeds = [k * 32 + 2 * VARIANT_SEED for k in range(Bb // 32)]
            with ctx.Pool(min(8, os.cpu_count() or 1)) as pool:            # (numpy only in the children)
                parts = pool.starmap(_synth_part, [(s0, 32, wl.fh) for s0 in seeds])
            cifs = np.concatenate([c for c, _ in parts]); cafs = np.concatenate([f for _, f in parts])
            cif_d, caf_d = torch.from_numpy(cifs).to(device), torch.from_numpy(cafs).to(device)
            variants = [(cifs, cafs, cif_d, caf_d)]
            wb = Workload.__new__(Workload)
            wb.__dict__.update(wl.__dict__)
            wb.B, wb.variants = Bb, variants
            wb.dec = native.CifCaf(wl.K, torch.from_numpy(wl.skeleton0))
            wb.host_out = torch.empty((Bb, wb.dec.max_annotations, wl.K, 4), dtype=torch.float32).pin_memory()
            wb.host_counts = torch.empty((Bb,), dtype=torch.int32).pin_memory()
            out = {'what': 'decode only: ONE call_batch of %d COCO-shaped images (%d synth batches of 32, seeds %d..), annotations copied '
                           'to the host; roofline: SURVEY 8d bytes of the 256-image launch over the dominant kernel\'s HIP-event time; '
                           'lanes: native.DecodeLanes with 256 images per call; b512: 512 images in one call (image queue)' % (Bb, Bb // 32, seeds[0])}
            rate = {}
            for n in (1, 2, 4):
                steps = 24
                leg = run_leg(wb, None, 'fp32', steps, 2 * n + 3, decode_only=True, n_streams=n)
                rate[n] = round(Bb * steps / leg['elapsed'], 1)
                if n == 1:
                    out['ms_per_batch_wall'] = round(leg['elapsed'] / steps * 1e3, 3)
                    out['annotations_per_batch'] = leg['n_ann']
            out['value'] = rate[1]
            out['lanes_images_per_s'] = {'one': rate[1], 'two': rate[2], 'four': rate[4]}
            alg = algorithmic_bytes(Bb, wl.K, wl.A, wl.fh, wl.fh, wl.stride, wb.dec.max_annotations)['decode_path']
            best = max(rate.values())
            out['best'] = {'images_per_s': best, 'GBps': round(alg * best / Bb / 1e9, 1), 'frac': round(alg * best / Bb / 1e9 / HBM_PEAK_GBPS, 5)}
            with torch.cuda.stream(dec_stream):
                out['roofline'] = decode_roofline(wb, variants, None, 6)
                out['parity'] = None if args.no_parity else parity_stamp(
                    lambda c, f: wb.dec.call_batch(c, wl.stride, f, wl.stride), variants, wl.skeleton0, wl.K)
                # 512 images in one call: the 256 twice
                c2, f2 = torch.cat([cif_d, cif_d]), torch.cat([caf_d, caf_d])
                for _ in range(2):
                    wb.dec.call_batch(c2, wl.stride, f2, wl.stride)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(8):
                    o2, i2, n2 = wb.dec.call_batch(c2, wl.stride, f2, wl.stride)
                torch.cuda.synchronize(device)
                ms = (time.perf_counter() - t0) / 8 * 1e3
                native.check_counts(n2)
                k512 = kernel_profile(Workload_view(wb, 2 * Bb, [(None, None, c2, f2)]), [(None, None, c2, f2)], None, 4)
                same = bool(torch.equal(n2[:Bb], n2[Bb:]))
                out['b512_one_call'] = {'images_per_s': round(2 * Bb / (ms * 1e-3), 1), 'ms_per_batch_wall': round(ms, 3),
                                        'kernels_ms': {k: round(v, 4) for k, v in k512.items()},
                                        'frac_dominant_kernel': round(2 * alg / (max(k512.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                                        'frac_decode_path_wall': round(2 * alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                                        'both_halves_same_counts': same}
                del c2, f2, o2
            wb.dec = None
            torch.cuda.empty_cache()
            return out
        guarded('decode_b256', b256_leg)

        # the adversarial case (BASELINE.md 3 ii): structureless all-active fields, what a random-init head
        # emits -- every cell passes every threshold.  Decode only, reported separately, with its own parity count.
        def all_active_leg():
            from openpifpaf_amd import synth
            pairs = [synth.adversarial_fields(500 + i, height=wl.fh, width=wl.fh, people=(3, 6, 1, 10)[i % 4] if wl.K == 17 else 0)
                     for i in range(wl.B)]
            cifs = np.stack([c for c, _ in pairs]); cafs = np.stack([f for _, f in pairs])
            cif_d, caf_d = torch.from_numpy(cifs).to(device), torch.from_numpy(cafs).to(device)
            variants = [(cifs, cafs, cif_d, caf_d)]
            # (every cell active: the CIF map reaches every tile -- a pool that holds the whole map, opa_shape::cifhr_pool_tiles)
            dec_full = native.CifCaf(wl.K, torch.from_numpy(wl.skeleton0), cifhr_pool_tiles='full')
            saved_dec, wl.dec = wl.dec, dec_full
            with torch.cuda.stream(dec_stream):
                for _ in range(2):
                    wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(10):
                    out, ids, counts = wl.dec.call_batch(cif_d, wl.stride, caf_d, wl.stride)
                torch.cuda.synchronize(device)
                ms = (time.perf_counter() - t0) / 10 * 1e3
                kernels = kernel_profile(wl, variants, None, 4)
                par = None if args.no_parity else parity_stamp(
                    lambda c, f: wl.dec.call_batch(c, wl.stride, f, wl.stride), variants, wl.skeleton0, wl.K)
            seeds = wl.dec.workspace_view('seed_count', torch.int32)[:wl.B].cpu().numpy()
            wl.dec = saved_dec
            cpu = None if args.no_cpu_baseline else cpu_baseline(cifs, cafs, wl.skeleton0, wl.K, 6.0)
            alg = algorithmic_bytes(wl.B, wl.K, wl.A, wl.fh, wl.fh, wl.stride, wl.dec.max_annotations)['decode_path']
            return {'what': 'decode only: %d all-active %dx%d field pairs (synth.adversarial_fields: sigmoid(N(0,1)) confidences, '
                            'every cell above every threshold) with 3 / 6 / 1 / 10 people planted into the noise, so that the '
                            'parity stamp compares poses, not a count of zero' % (wl.B, wl.fh, wl.fh),
                    'cpu_baseline': cpu,
                    'decode_vs_cpu_1thread': round(wl.B / (ms * 1e-3) / cpu['value'], 1) if cpu else None,
                    'decode_only_images_per_s': round(wl.B / (ms * 1e-3), 1), 'ms_per_batch_wall': round(ms, 3),
                    'decode_ms': round(sum(kernels.values()), 4),
                    'kernels_ms': {k: round(v, 4) for k, v in kernels.items()},
                    'seeds_per_image': {'mean': float(seeds.mean()), 'max': int(seeds.max())},
                    'roofline': {'frac': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
                    'parity': par}
        guarded('all_active', all_active_leg)


        # BASELINE.json configs[0], the plumbing baseline: resnet18, ONE 321x321 image, the reference's CPU-only predict
        # (network on the host cores in float32 -> its C++ CifCaf decoder on one thread) next to this path (network on the
        # MI355X -> HIP decode -> annotations on the host), on the same 41x41 synthetic fields
        def r0_leg():
            from openpifpaf_amd import synth
            edge = 321
            w0 = Workload(2, 1, 0, device, edge, 2, backbone='resnet18')
            fh = (edge - 1) // w0.stride + 1
            pairs = [synth.synth_fields(900 + i, (2, 5)[i % 2], height=fh, width=fh) for i in range(4)]
            w0.variants = [(c[None], f[None], torch.from_numpy(c[None]).to(device), torch.from_numpy(f[None]).to(device)) for c, f in pairs]
            img = torch.randn((1, 3, edge, edge), generator=torch.Generator().manual_seed(11))
            model = build_model(w0, 'fp32')
            img_d = img.to(device).contiguous(memory_format=torch.channels_last)
            n = 30
            def ours(i):
                _, _, cd, fd = w0.variants[i % 4]
                with torch.no_grad():
                    model(img_d)
                out, ids, counts = w0.dec.call_batch(cd, w0.stride, fd, w0.stride)
                return out.cpu(), counts.cpu()
            for i in range(5):
                ours(i)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(n):
                ours(i)
            ours_ms = (time.perf_counter() - t0) / n * 1e3
            out = {'what': 'BASELINE configs[0]: resnet18, one %dx%d image (fields %dx%d); ours = network on the MI355X + HIP decode + '
                           'D2H of the annotations, eager; reference = the same float32 network on the HOST cores + the reference C++ '
                           'decoder on one thread (its CPU-only predict)' % (edge, edge, fh, fh),
                   'value': round(1e3 / ours_ms, 1), 'ms_per_step': round(ours_ms, 3), 'eager_ms_per_image': round(ours_ms, 3)}
            if not args.no_cpu_baseline:
                kind, make, torch_ = reference_decoder(w0.skeleton0, w0.K, None)
                decode = make(True)
                from openpifpaf_amd import network
                cpu_model = network.factory(w0.backbone, [w0.cif_meta, w0.caf_meta]).eval()   # plain PyTorch modules, float32, host cores
                threads0 = torch.get_num_threads()
                with torch.no_grad():
                    cpu_model(img)                                       # warm-up
                    t0 = time.perf_counter()
                    k = 0
                    while time.perf_counter() - t0 < 4.0 or k < 3:
                        cpu_model(img)
                        k += 1
                    nn_ms = (time.perf_counter() - t0) / k * 1e3
                t0 = time.perf_counter()
                k = 0
                while time.perf_counter() - t0 < 2.0 or k < 3:
                    decode(pairs[k % 4][0], pairs[k % 4][1])
                    k += 1
                dec_ms = (time.perf_counter() - t0) / k * 1e3
                out['reference_cpu'] = {'kind': kind, 'network_ms': round(nn_ms, 2), 'network_threads': threads0,
                                        'decode_ms_1thread': round(dec_ms, 3), 'ms_per_image': round(nn_ms + dec_ms, 2),
                                        'images_per_s': round(1e3 / (nn_ms + dec_ms), 2)}
                out['vs_reference_cpu'] = round((nn_ms + dec_ms) / ours_ms, 2)
            with torch.cuda.stream(dec_stream):
                out['parity'] = None if args.no_parity else parity_stamp(
                    lambda c, f: w0.dec.call_batch(c, w0.stride, f, w0.stride), w0.variants, w0.skeleton0, w0.K)
            return out
        guarded('config1_resnet18_321', r0_leg)

        # the literal configs[1]: batch 1
        def batch1_leg():
            w1 = Workload(2, 1, 0, device, args.long_edge, 2)
            w1.variants = [(c[i:i + 1], f[i:i + 1], cd[i:i + 1].contiguous(), fd[i:i + 1].contiguous())
                           for (c, f, cd, fd) in wl.variants for i in (3, 2)]      # 20-, 10-, ... person images
            img1 = torch.randn((1, 3, args.long_edge, args.long_edge), generator=torch.Generator().manual_seed(7)).to(device)
            img1 = img1.contiguous(memory_format=torch.channels_last)
            model = build_model(w1, 'fp32')
            out = {}
            n = 30

            def one_eager(i):
                _, _, cif_d, caf_d = w1.variants[i % len(w1.variants)]
                with torch.no_grad():
                    model(img1)
                o, ids, counts = w1.dec.call_batch(cif_d, w1.stride, caf_d, w1.stride)
                w1.host_out.copy_(o, non_blocking=True)
                w1.host_counts.copy_(counts, non_blocking=True)
                torch.cuda.synchronize(device)
            for i in range(5):
                one_eager(i)
            t0 = time.perf_counter()
            for i in range(n):
                one_eager(i)
            out['eager_ms_per_image'] = round((time.perf_counter() - t0) / n * 1e3, 3)
            # network alone / decode alone at batch 1
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(n):
                with torch.no_grad():
                    model(img1)
            torch.cuda.synchronize(device)
            out['network_ms_per_image'] = round((time.perf_counter() - t0) / n * 1e3, 3)
            t0 = time.perf_counter()
            for i in range(n):
                _, _, cif_d, caf_d = w1.variants[i % len(w1.variants)]
                w1.dec.call_batch(cif_d, w1.stride, caf_d, w1.stride)
                torch.cuda.synchronize(device)
            out['decode_ms_per_image_eager'] = round((time.perf_counter() - t0) / n * 1e3, 3)
            out['decode_kernels_ms'] = {k: round(v, 4) for k, v in kernel_profile(w1, w1.variants, None, 8).items()}
            # where an eager step's host time goes: queueing the network (Python + launches, no wait), queueing the decode,
            # then waiting for the device
            tq_net = tq_dec = tq_sync = 0.0
            for i in range(n):
                _, _, cif_d, caf_d = w1.variants[i % len(w1.variants)]
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                with torch.no_grad():
                    model(img1)
                t1 = time.perf_counter()
                o, ids, counts = w1.dec.call_batch(cif_d, w1.stride, caf_d, w1.stride)
                w1.host_out.copy_(o, non_blocking=True)
                w1.host_counts.copy_(counts, non_blocking=True)
                t2 = time.perf_counter()
                torch.cuda.synchronize(device)
                t3 = time.perf_counter()
                tq_net += t1 - t0; tq_dec += t2 - t1; tq_sync += t3 - t2
            out['eager_breakdown_ms'] = {'host_queues_network': round(tq_net / n * 1e3, 3),
                                         'host_queues_decode_and_copies': round(tq_dec / n * 1e3, 3),
                                         'host_waits_for_device': round(tq_sync / n * 1e3, 3),
                                         'device_network': out['network_ms_per_image'],
                                         'device_decode': round(sum(out['decode_kernels_ms'].values()), 3)}
            # the whole step as ONE HIP graph: network + decode captured together, fields refilled in place
            try:
                st = torch.cuda.Stream()
                cif_s, caf_s = w1.variants[0][2].clone(), w1.variants[0][3].clone()
                gdec = native.CifCaf(w1.K, torch.from_numpy(w1.skeleton0))
                with torch.cuda.stream(st):
                    with torch.no_grad():
                        model(img1)
                    gdec.call_batch(cif_s, w1.stride, caf_s, w1.stride)
                st.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=st):
                    with torch.no_grad():
                        model(img1)
                    g_out, g_ids, g_counts = gdec.call_batch(cif_s, w1.stride, caf_s, w1.stride)
                gdec._pinned = True

                def one_graph(i):
                    _, _, cif_d, caf_d = w1.variants[i % len(w1.variants)]
                    with torch.cuda.stream(st):
                        cif_s.copy_(cif_d, non_blocking=True)
                        caf_s.copy_(caf_d, non_blocking=True)
                        graph.replay()
                        w1.host_out.copy_(g_out, non_blocking=True)
                        w1.host_counts.copy_(g_counts, non_blocking=True)
                    st.synchronize()
                for i in range(5):
                    one_graph(i)
                t0 = time.perf_counter()
                for i in range(n):
                    one_graph(i)
                out['hip_graph_ms_per_image'] = round((time.perf_counter() - t0) / n * 1e3, 3)
                native.check_counts(w1.host_counts)
            except Exception as e:
                out['hip_graph_ms_per_image'] = None
                out['hip_graph_error'] = repr(e)
            # the reference flow at batch 1, run: network -> .cpu() -> reference decoder (1 thread, reused instance)
            if not args.no_cpu_baseline:
                kind, make, _ = reference_decoder(w1.skeleton0, w1.K)
                decode = make(True)
                with torch.no_grad():
                    [h.cpu() for h in model(img1)]
                t0 = time.perf_counter()
                for i in range(n):
                    with torch.no_grad():
                        [h.cpu() for h in model(img1)]
                    c, f = w1.variants[i % len(w1.variants)][:2]
                    decode(c[0], f[0])
                out['reference_flow_ms_per_image'] = round((time.perf_counter() - t0) / n * 1e3, 3)
                out['vs_reference_flow'] = {'eager': round(out['reference_flow_ms_per_image'] / out['eager_ms_per_image'], 2),
                                            'hip_graph': round(out['reference_flow_ms_per_image'] / out['hip_graph_ms_per_image'], 2)
                                            if out.get('hip_graph_ms_per_image') else None}
            if not args.no_parity:
                out['parity'] = parity_stamp(lambda c, f: w1.dec.call_batch(c, w1.stride, f, w1.stride), w1.variants,
                                             w1.skeleton0, w1.K)
            out['what'] = ('BASELINE configs[1] literally: resnet50 641x641, batch 1 (reference predictor.py:15), float32; latency '
                           'per image incl. the host copy of the annotations and a device synchronise; fields: the 20- and '
                           '10-person images of the headline\'s two field batches in turn')
            return out
        guarded('batch1', batch1_leg)

        for cid in (3, 4):
            def cfg_leg(cid=cid):
                w = Workload(cid, CONFIGS[cid]['batch'], 0, device, args.long_edge, 2)
                gi = torch.Generator(device='cpu').manual_seed(1000)
                im = torch.randn((w.B, 3, args.long_edge, args.long_edge), generator=gi).to(device)
                im = im.contiguous(memory_format=torch.channels_last)
                r = full_config(w, im, args.extra_steps, 2, primary='fp32', bf16_leg=False,
                                cpu_seconds=5.0 if cid == 3 else 8.0, profile_steps=4, params=None, reference_steps=1)
                r['steps'] = args.extra_steps
                return r
            guarded('config%d' % cid, cfg_leg)
            torch.cuda.empty_cache()
        if line is not None:
            line['configs'] = others
    if rank == 0:
        line['bench_seconds'] = round(time.perf_counter() - t_program, 1)
        write_detail(line)
        sys.stderr.flush()
        print(compact_line(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return line


if __name__ == '__main__':
    main()
