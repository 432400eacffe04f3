"""Round-3 probe: per-kernel HIP-event times and the association kernel's always-on statistics for one of the
BASELINE decode shapes.

    python tools/gpu/r3_probe.py [--config coco|wholebody] [--fc] [--batch N] [--alternate] [--trace IMAGE] [--check]

--fc         the reference benchmark CLI's setting (--force-complete-pose + zero thresholds, benchmark.py:77-79)
--alternate  decode two different field batches in turn (lazy tile clear / caches see changing input)
--check      compare every image with the reference decoder (oracle/_ref) or the restatement
"""
import argparse
import os
os.environ.setdefault('OPA_ASSOC_TIMING', '1')     # the coordinator's per-phase tick counters are off by default (read once, when the library is loaded)
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpifpaf_amd import _lib, constants, native, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='coco', choices=('coco', 'wholebody'))
ap.add_argument('--fc', action='store_true')
ap.add_argument('--batch', type=int, default=None)
ap.add_argument('--alternate', action='store_true')
ap.add_argument('--trace', type=int, default=None)
ap.add_argument('--check', action='store_true')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--bench-batches', action='store_true', help='with --alternate: the two field batches bench.py alternates (seed0 0 and 100000) instead of 0 and 1000')
ap.add_argument('--flush', action='store_true', help='overwrite 1 GB before every timed call: the fields come from HBM, not the 256 MB Infinity Cache')
ap.add_argument('--param', action='append', default=[], metavar='NAME=VALUE',
                help='decoder parameter override (sensitivity experiments), e.g. --param reverse_match=0 --param greedy=1')
ap.add_argument('--debug', action='append', default=[], metavar='NAME=VALUE', help='opa_debug switch of the decoder, e.g. --debug scored_one_pass=0')
args = ap.parse_args()

if args.config == 'wholebody':
    wb = constants.wholebody()
    skel1, K, people, pose, B = wb['skeleton'], 133, (1, 3, 6, 10), wb['standing_pose'], args.batch or 16
    kw = dict(people=people, pose=pose, skeleton=skel1)
else:
    skel1, K, people, B = constants.COCO_PERSON_SKELETON, 17, synth.PEOPLE_CYCLE, args.batch or 32
    kw = {}
skel0 = np.asarray(skel1, dtype=np.int64) - 1
fc_kw = dict(force_complete=1, keypoint_threshold=0.0, keypoint_threshold_rel=0.0,
             nms_instance_threshold=0.0, nms_keypoint_threshold=0.0)
overrides = dict(fc_kw) if args.fc else {}
for kv in args.param:
    name, value = kv.split('=', 1)
    overrides[name] = float(value) if '.' in value else int(value)
params = _lib.default_params(**overrides) if overrides else None

batches = []
for s in (((0, 100000) if args.bench_batches else (0, 1000)) if args.alternate else (0,)):
    cifs, cafs = synth.synth_batch(B, seed0=s, **kw)
    batches.append((cifs, cafs, torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()))
dec = native.CifCaf(K, torch.from_numpy(skel0))
if args.debug:
    dec.set_debug(**{kv.split('=', 1)[0]: (float(kv.split('=', 1)[1]) if '.' in kv else int(kv.split('=', 1)[1])) for kv in args.debug})
for i in range(3):
    _, _, cd, fd = batches[i % len(batches)]
    out, ids, counts = dec.call_batch(cd, 8, fd, 8, params=params)
torch.cuda.synchronize()
print('config %s fc=%s batch %d alternate=%s workspace %.2f GB' % (args.config, args.fc, B, args.alternate,
                                                                  dec._last[1].numel() / 1e9))

acc = {}
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda') if args.flush else None
for i in range(args.reps):
    _, _, cd, fd = batches[i % len(batches)]
    if flush is not None:
        flush.fill_(float(i))
        torch.cuda.synchronize()
    _lib.profile_begin(native._stream())
    out, ids, counts = dec.call_batch(cd, 8, fd, 8, params=params)
    for name, ms in _lib.profile_end():
        acc.setdefault(name, []).append(ms)
tot = sum(np.mean(v) for v in acc.values())
print('  '.join('%s %.1f us' % (k.replace('_kernel', ''), 1e3 * np.mean(v)) for k, v in acc.items()),
      ' | decode %.3f ms -> %.0f images/s' % (tot, B / tot * 1e3))
t0 = time.perf_counter()
for i in range(20):
    _, _, cd, fd = batches[i % len(batches)]
    dec.call_batch(cd, 8, fd, 8, params=params)
torch.cuda.synchronize()
print('wall: %.3f ms per call_batch (20 back-to-back calls, one stream)' % ((time.perf_counter() - t0) / 20 * 1e3))

# statistics of the LAST call (batch (reps-1) % len)
cifs, cafs, cd, fd = batches[(20 - 1) % len(batches)]
out, ids, counts = dec.call_batch(cd, 8, fd, 8, params=params)
torch.cuda.synchronize()
st = dec.assoc_stats().cpu().numpy()
status = dec.workspace_view('status', torch.int32)[:B].cpu().numpy()
print('status', status.tolist())
print('img poses seeds | started accepted stopped dropped pre-stop mispred refills | growth-us total-us iters '
      'head-wait-us grower-busy-us scans us/scan growers')
order = np.argsort(-st[:, 9])[:8]
for b in order:
    s = st[b]
    print('%3d %5d %5d | %7d %8d %9d %7d %8d %7d %7d | %9.0f %8.0f %6.0f %13.0f %14.0f %5d %7.2f %7d' % (
        b, native.count_rows(int(counts[b])), s[7], s[0], s[1], s[2], s[3], s[4], s[5], s[6],
        s[8] / 100, s[9] / 100, s[15], s[12] / 100, s[10] / 100, s[11], s[10] / 100 / max(1, s[11]), s[13]))
print('coordinator us per image (wait-iters | head-wait commit refill hand-out other | refill waiting for marks | dup-dropped):')
for b in order:
    s = st[b]
    print('%3d  %5d | %6.0f %6.0f %6.0f %6.0f %6.0f | %6.0f | %6d' % (
        b, s[16], s[12] / 100, s[17] / 100, s[18] / 100, s[19] / 100, (s[8] - s[12] - s[17] - s[18] - s[19]) / 100,
        s[20] / 100, s[23]))
tot_s = st.sum(axis=0)
print('batch: started %d accepted %d cancelled %d dropped %d; slowest image %.0f us, mean %.0f us; lists: max %d mean %.0f' % (
    tot_s[0], tot_s[1], tot_s[2], tot_s[3], st[:, 9].max() / 100, st[:, 9].mean() / 100,
    int(dec.workspace_view('list_counts', torch.int32)[:B * len(skel0) * 2].max()),
    float(dec.workspace_view('list_counts', torch.int32)[:B * len(skel0) * 2].float().mean())))
print('level walk: connection values from the memo %d, evaluated on demand %d' % (tot_s[21], tot_s[22]))
if args.fc:
    lc = dec.workspace_view('list_counts_fc', torch.int32)[:B * len(skel0) * 2]
    print('force-complete lists: max %d mean %.0f' % (int(lc.max()), float(lc.float().mean())))
if args.trace is not None:
    b = args.trace
    tr = dec.workspace_view('assoc_trace', torch.int32).view(B, 64, 4)[b].cpu().numpy()
    n = int(st[b][1])
    print('image %d: commit#  seed  grower  handed-out-us  done-us  commit-us  (growth us, waited-for-growth us)' % b)
    prev = 0
    for k in range(min(n, 64)):
        tc, te, td, sg = [int(v) for v in tr[k]]
        print('  %3d %6d %3d %9.1f %9.1f %9.1f   growth %6.1f  head waited %6.1f' % (
            k, sg & 0xFFFFFF, sg >> 24, te / 100, td / 100, tc / 100, (td - te) / 100, max(0, td - prev) / 100))
        prev = tc
if args.check:
    from oracle import port, reference
    pp = port.default_params(**fc_kw) if args.fc else None
    use_ref = reference.available()
    if use_ref:
        reference.load().set_num_threads(1)
        reference.reset_statics()
        if pp is not None:
            reference.apply_params(pp)
    worst, t0, n = 0.0, time.perf_counter(), 0
    for b in range(B):
        if use_ref:
            r = reference.decode(cifs[b], 8, cafs[b], 8, skel0)[0]
        else:
            r = port.decode(cifs[b], 8, cafs[b], 8, skel0, params=pp)[0]
        g = out[b, :native.count_rows(int(counts[b]))].cpu().numpy()
        assert g.shape == r.shape, (b, g.shape, r.shape)
        if g.size:
            assert np.array_equal(g[..., 0] > 0, r[..., 0] > 0), b
            worst = max(worst, float(np.abs(g.astype(np.float64) - r).max()))
        n += 1
    if use_ref:
        reference.reset_statics()
    dt = time.perf_counter() - t0
    assert worst <= 1e-4, worst
    print('parity vs %s: %d images, max |delta| %.3g; CPU %.1f ms/image (1 thread)' % (
        'the reference' if use_ref else 'the restatement', n, worst, dt / n * 1e3))
