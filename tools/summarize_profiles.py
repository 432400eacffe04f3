"""Turns rocprofv3 CSV output (tools/collect_profiles.sh) into the markdown summary
committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(root, sub, '**', pattern), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.split('(')[0]
    if name.startswith('void opa::'):
        name = name[5:]
    return name if len(name) < 90 else name[:87] + '...'


def kernel_stats(sub, title, top=25):
    path = find(sub, '*kernel_stats.csv')
    print('## %s\n' % title)
    if not path:
        print('(no kernel_stats.csv found)\n')
        return
    rows = list(csv.DictReader(open(path)))
    print('| kernel | calls | total ms | avg us | % |')
    print('|---|---|---|---|---|')
    for r in rows[:top]:
        print('| `%s` | %s | %.3f | %.2f | %.2f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    print()


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
    if len(idx) < 8:
        print('(too few steps in the trace)\n')
        return
    a, b = idx[-7], idx[-6]
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


def pmc(sub, counter):
    path = find(sub, '*counter_collection.csv')
    if not path:
        return {}
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        k = short(r['Kernel_Name'])
        acc[k].append(float(r['Counter_Value']))
    # median over the launches: the first call on a workspace writes the whole CifHr map (lazy clear not
    # primed yet) and would dominate a mean
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items() if v}


print('# rocprofv3 summary (round 2)\n')
print('Commands: see tools/collect_profiles.sh.  Batch 32 per launch, COCO-17 fields 81x81, stride 8.\n')
steady_state_step('bench', 'bench.py headline leg (float32 network + decode): ONE steady-state step, batch 32')
steady_state_step('bench_bf16', 'bench.py --backbone-dtype bf16 (bfloat16 network + decode): ONE steady-state step, batch 32')
kernel_stats('decode', 'bench.py --decode-only --steps 10 --warmup 2, kernel trace')
fetch, write = pmc('pmc_FETCH_SIZE', 'FETCH_SIZE'), pmc('pmc_WRITE_SIZE', 'WRITE_SIZE')
print('## HBM traffic counters per launch (decode only)\n')
print('FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the TCC_EA request counters; on gfx950 '
      'FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): the "corrected" column '
      'doubles it.\n')
print('| kernel | FETCH_SIZE (KB) | x2 corrected (MB) | WRITE_SIZE (KB) | write (MB) |')
print('|---|---|---|---|---|')
for k in sorted(set(fetch) | set(write)):
    if not k.startswith('opa::') and 'fillBuffer' not in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    print('| `%s` | %.1f | %.2f | %.1f | %.2f |' % (k, f, 2 * f * 1024 / 1e6, w, w * 1024 / 1e6))

# cache / LDS / instruction mix of the decode kernels
hit, miss = pmc('pmc_TCC_HIT_sum', 'TCC_HIT_sum'), pmc('pmc_TCC_HIT_sum', 'TCC_MISS_sum')
conf, lds_act, lds_inst = (pmc('pmc_SQ_LDS_BANK_CONFLICT', c) for c in ('SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_LDS'))
valu, salu, wcyc = (pmc('pmc_SQ_INSTS_VALU', c) for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES'))
if hit or conf or valu:
    print('\n## L2 hit rate, LDS bank conflicts, instruction mix (decode only, per launch, median)\n')
    print('L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS '
          '(cycles); instructions are per launch, summed over all waves.\n')
    print('| kernel | L2 hit rate | LDS bank-conflict cycles / LDS active cycles | LDS instr | VALU instr | SALU instr | wave cycles |')
    print('|---|---|---|---|---|---|---|')
    for k in sorted(set(hit) | set(conf) | set(valu)):
        if not k.startswith('opa::'):
            continue
        h, m = hit.get(k, 0.0), miss.get(k, 0.0)
        print('| `%s` | %s | %s | %.3g | %.3g | %.3g | %.3g |' % (
            k, '%.1f %%' % (100 * h / (h + m)) if h + m else '-',
            '%.1f %%' % (100 * conf.get(k, 0.0) / lds_act[k]) if lds_act.get(k) else '-',
            lds_inst.get(k, 0.0), valu.get(k, 0.0), salu.get(k, 0.0), wcyc.get(k, 0.0)))

# machine-readable PMC traffic for bench.py's roofline.traffic
import json
import hashlib
_h = hashlib.sha256()
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpifpaf_amd', 'csrc')
for _n in sorted(os.listdir(_csrc)):
    if _n.endswith(('.hip', '.hpp')):
        _h.update(open(os.path.join(_csrc, _n), 'rb').read())
out = {'batch': 32, 'config': 2, 'kernel_source_hash': _h.hexdigest()[:16], 'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --decode-only, per launch; '
       'units KiB of TCC_EA requests; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 x2 read correction '
       'of MI355X_MICROARCH.md (calibrated for 16-B/lane streams only; 4-B/lane reads are uncalibrated)', 'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith('opa::'):
        continue
    name = k.split('::')[1].split('<')[0]
    out['kernels'][name] = {'FETCH_SIZE': round(fetch.get(k, 0.0), 1), 'WRITE_SIZE': round(write.get(k, 0.0), 1),
                            'hbm_bytes': int((2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)}
if out['kernels']:
    json.dump(out, open(os.path.join(root, 'pmc_traffic.json'), 'w'), indent=1)
