#!/bin/bash
# round 6: kernel timeline (start / end per dispatch, per stream) of the last decodes of a probe run: do the two branches overlap?
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"; OUT="$REPO/gpurun_out/r6/${TAG:-trace}"; mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH="$REPO"
cd /tmp; rm -rf /tmp/trace_r6
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_r6 -o tr -- \
    python "$REPO/tools/gpu/r3_probe.py" --config coco --batch ${BATCH:-256} --alternate --reps 2 > "$OUT/stdout.log" 2> "$OUT/stderr.log"
f=$(find /tmp/trace_r6 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee "$OUT/timeline.log"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'opa::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-30:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    name = r['Kernel_Name'].split('(')[0].replace('opa::', '').replace('void ', '')[:40]
    print('%-40s queue %-6s start %9.1f us  end %9.1f us  dur %8.1f' % (name, r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
