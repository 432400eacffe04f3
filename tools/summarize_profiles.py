"""Turns rocprofv3 CSV output (tools/collect_profiles.sh) into the markdown summary and the machine-readable PMC
traffic file committed under profiles/r3/.

    python tools/summarize_profiles.py <output root of collect_profiles.sh>

Layout of the root: ``<workload>/stats`` (kernel trace + stats of tools/gpu/r3_probe.py on that workload),
``<workload>/pmc_<COUNTER...>`` (one rocprofv3 --pmc pass each), ``bench`` / ``bench_bf16`` (kernel trace of bench.py's
headline / bfloat16 leg).  Workloads: config2 (COCO-17 batch 32), config2_fc, config4 (wholebody batch 16), config4_fc.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
WORKLOADS = [('config2', 'COCO-17 (configs 2 / 3 fields), batch 32, default flags', 2, 32, False),
             ('config2_fc', 'COCO-17, batch 32, the reference benchmark\'s force-complete setting', 2, 32, True),
             ('config4', 'wholebody 133 keypoints / 160 bones (config 4), batch 16, default flags', 4, 16, False),
             ('config4_fc', 'wholebody, batch 16, force complete', 4, 16, True),
             ('config2_b256', 'COCO-17, 256 images in ONE call_batch (round 6)', 2, 256, False)]
DECODE_KERNELS = ('zero_kernel', 'cif_active_kernel', 'cifhr_tile_kernel', 'cifhr_worktile_kernel', 'tile_state_roll_kernel',
                  'cifseeds_fill_kernel', 'cifseeds_fill_cand_kernel', 'cifseeds_sort2k_kernel', 'cifseeds_sort_kernel',
                  'cifseeds_sort_scored_kernel', 'cifseeds_rankmerge_kernel', 'cifseeds_tie_kernel', 'cafscored_kernel',
                  'cafscored2_kernel', 'assoc_order_kernel', 'cifcaf_assoc_kernel', 'cifcaf_fc_kernel')


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(root, sub, '**', pattern), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.split('(')[0]
    if name.startswith('void opa::'):
        name = name[5:]
    return name if len(name) < 90 else name[:87] + '...'


def base(name):
    """opa::cifcaf_assoc_kernel<true, 12> -> cifcaf_assoc_kernel"""
    s = short(name)
    return s.split('::')[-1].split('<')[0] if s.startswith('opa::') else s


def kernel_stats(sub, title, top=14):
    path = find(sub, '*kernel_stats.csv')
    print('### %s\n' % title)
    if not path:
        print('(no kernel_stats.csv found)\n')
        return {}
    rows = list(csv.DictReader(open(path)))
    print('| kernel | calls | total ms | avg us | % |')
    print('|---|---|---|---|---|')
    out = {}
    for r in rows[:top]:
        print('| `%s` | %s | %.3f | %.2f | %.2f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
        out[base(r['Name'])] = float(r['AverageNs']) / 1e3
    print()
    return out


def steady_state_step(sub, title, top=16):
    """MIOpen's find mode benchmarks many candidate solvers during warm-up, which pollutes the
    whole-run statistics; this isolates ONE timed step: the kernels between two consecutive
    association-kernel launches near the end of the trace."""
    path = find(sub, '*kernel_trace.csv')
    print('## %s\n' % title)
    if not path:
        print('(no kernel_trace.csv found)\n')
        return
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'cifcaf_assoc' in r['Kernel_Name']]
    # a timed step = the stretch between two consecutive association launches that holds a whole network forward (the
    # run ends with decode-only calls: profile steps, tile count, parity stamp); take the last but one such stretch
    pairs = [(a, b) for a, b in zip(idx, idx[1:]) if b - a > 40]
    if len(pairs) < 3:
        print('(too few steps in the trace)\n')
        return
    a, b = pairs[-2]
    seg = rows[a + 1:b + 1]
    agg = defaultdict(lambda: [0, 0.0])
    for r in seg:
        k = short(r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    total = sum(v[1] for v in agg.values())
    wall = (max(int(r['End_Timestamp']) for r in seg) - min(int(r['Start_Timestamp']) for r in seg)) / 1e6
    print('%d kernels in the step, sum of kernel time %.2f ms, wall %.2f ms\n' % (len(seg), total / 1e3, wall))
    print('| kernel | launches | total ms | avg us |')
    print('|---|---|---|---|')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('| `%s` | %d | %.3f | %.1f |' % (k, n, t / 1e3, t / n))
    print()


def pmc_per_decode(sub, counter, last_calls=4):
    """Counter value per kernel and per DECODE CALL: the sum over the kernel's launches inside the last `last_calls`
    decode calls of the run / last_calls (a call = everything up to and including its last association-stage kernel;
    the first call on a workspace writes the whole CifHr map, and the probe alternates two field batches)."""
    path = find(sub, '*counter_collection.csv')
    if not path:
        return {}
    rows = [r for r in csv.DictReader(open(path)) if r.get('Counter_Name') == counter]
    if not rows:
        return {}
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    names = [base(r['Kernel_Name']) for r in rows]
    last_stage = 'cifcaf_fc_kernel' if 'cifcaf_fc_kernel' in names else 'cifcaf_assoc_kernel'
    ends = [i for i, n in enumerate(names) if n == last_stage]
    if len(ends) < last_calls + 1:
        return {}
    lo, hi = ends[-last_calls - 1] + 1, ends[-1] + 1
    acc = defaultdict(float)
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        acc[n] += float(r['Counter_Value'])
    return {k: v / last_calls for k, v in acc.items()}


ONLY_STEPS = '--only-steps' in sys.argv
print('# rocprofv3 summary (round %s)%s\n' % (os.environ.get('ROUND', 'r6').lstrip('r'), ': one steady-state step of bench.py' if ONLY_STEPS else ''))
print('Commands: tools/collect_profiles.sh (every pass: `rocprofv3 ... -- python tools/gpu/r3_probe.py --config ... --alternate`,'
      ' i.e. two different field batches decoded in turn; counters in their own passes with --kernel-trace only).\n')
steady_state_step('bench', 'bench.py headline leg (float32 network + decode): ONE steady-state step, batch 32')
steady_state_step('bench_bf16', 'bench.py --backbone-dtype bf16 (bfloat16 network + decode): ONE steady-state step, batch 32')

if ONLY_STEPS:
    sys.exit(0)
print('## Decode kernels per workload (kernel trace, averages over the run)\n')
stats = {}
for key, title, cfg, B, fc in WORKLOADS:
    stats[key] = kernel_stats(os.path.join(key, 'stats'), '%s -- %s' % (key, title))

# (the hash of the DECODE kernels' sources, as bench.kernel_source_hash computes it: producer-side files do not count)
PRODUCER_SOURCES = ('dwconv.hip', 'epilogue.hip', 'gemm_epilogue.hip', 'gemm_f32.hip', 'gemm_f32x3.hip', 'head.hip', 'winograd.hip')
h = hashlib.sha256()
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpifpaf_amd', 'csrc')
for n in sorted(os.listdir(csrc)):
    if n.endswith(('.hip', '.hpp')) and n not in PRODUCER_SOURCES:
        h.update(open(os.path.join(csrc, n), 'rb').read())
out = {'kernel_source_hash': h.hexdigest()[:16],
       'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/gpu/r3_probe.py --alternate, per decode call '
               '(sum over the launches of a kernel inside one call, mean of the last 4 calls); units KiB of TCC_EA requests; '
               'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 x2 read correction of MI355X_MICROARCH.md '
               '(calibrated for 16-B/lane streams; 4-B/lane reads are uncalibrated)',
       'workloads': {}}
print('## HBM traffic counters per decode call\n')
print('FETCH_SIZE / WRITE_SIZE are reported in KiB of the L2\'s memory-side requests; on gfx950 FETCH_SIZE counts a wide '
      'coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section): "read MB" doubles it.  Infinity-Cache hits are '
      'counted as traffic.  Algorithmic bytes = SURVEY 8d (CIF + CAF + annotations) x images per call.\n')
for key, title, cfg, B, fc in WORKLOADS:
    fetch = pmc_per_decode(os.path.join(key, 'pmc_FETCH_SIZE'), 'FETCH_SIZE')
    write = pmc_per_decode(os.path.join(key, 'pmc_WRITE_SIZE'), 'WRITE_SIZE')
    if not fetch and not write:
        continue
    print('### %s\n' % key)
    print('| kernel | FETCH_SIZE (KiB) | read MB (x2) | WRITE_SIZE (KiB) | write MB | HBM MB |')
    print('|---|---|---|---|---|---|')
    entry = {'kernels': {}, 'batch': B, 'config': cfg, 'force_complete': fc}
    total = 0
    for k in DECODE_KERNELS:
        if k not in fetch and k not in write:
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        hbm = int((2 * f + w) * 1024)
        total += hbm
        entry['kernels'][k] = {'FETCH_SIZE': round(f, 1), 'WRITE_SIZE': round(w, 1), 'hbm_bytes': hbm}
        print('| `%s` | %.1f | %.2f | %.1f | %.2f | %.2f |' % (k, f, 2 * f * 1024 / 1e6, w, w * 1024 / 1e6, hbm / 1e6))
    K, A = (17, 19) if cfg != 4 else (133, 160)
    alg = B * (K * 5 * 6561 * 4 + A * 8 * 6561 * 4 + 128 * K * 4 * 4)
    entry['decode_path_hbm_bytes'] = total
    entry['algorithmic_bytes'] = alg
    print('| **decode path** | | | | | **%.1f** = %.2f x the %.1f MB of SURVEY 8d |\n' % (total / 1e6, total / alg, alg / 1e6))
    out['workloads']['config%d%s_batch%d' % (cfg, '_fc' if fc else '', B)] = entry

print('## L2 hit rate, LDS bank conflicts, instruction mix (per decode call)\n')
print('L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS '
      '(cycles); instructions summed over all waves.\n')
for key, title, cfg, B, fc in WORKLOADS:
    hit = pmc_per_decode(os.path.join(key, 'pmc_TCC_HIT_sum'), 'TCC_HIT_sum')
    miss = pmc_per_decode(os.path.join(key, 'pmc_TCC_HIT_sum'), 'TCC_MISS_sum')
    conf, act, inst = (pmc_per_decode(os.path.join(key, 'pmc_SQ_LDS_BANK_CONFLICT'), c)
                       for c in ('SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_LDS'))
    valu, salu, wcyc = (pmc_per_decode(os.path.join(key, 'pmc_SQ_INSTS_VALU'), c)
                        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES'))
    if not (hit or conf or valu):
        continue
    print('### %s\n' % key)
    print('| kernel | L2 hit rate | LDS bank-conflict cycles / LDS active cycles | LDS instr | VALU instr | SALU instr | wave cycles |')
    print('|---|---|---|---|---|---|---|')
    for k in DECODE_KERNELS:
        if k not in hit and k not in conf and k not in valu:
            continue
        hh, mm = hit.get(k, 0.0), miss.get(k, 0.0)
        print('| `%s` | %s | %s | %.3g | %.3g | %.3g | %.3g |' % (
            k, '%.1f %%' % (100 * hh / (hh + mm)) if hh + mm else '-',
            '%.1f %%' % (100 * conf.get(k, 0.0) / act[k]) if act.get(k) else '-',
            inst.get(k, 0.0), valu.get(k, 0.0), salu.get(k, 0.0), wcyc.get(k, 0.0)))
    print()

if out['workloads']:
    json.dump(out, open(os.path.join(root, 'pmc_traffic.json'), 'w'), indent=1)
