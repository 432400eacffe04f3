#!/bin/bash
# per-kernel durations of the decode path from a rocprofv3 kernel trace (decode only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT="$PWD/gpurun_out/decode_trace"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o decode -- \
    python "$OLDPWD/bench.py" --decode-only --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stdout.log" 2> "$OUT/stderr.log"
cd "$OLDPWD"
python - <<'PY'
import csv, glob
p = glob.glob('gpurun_out/decode_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:14]:
    print('%-60s calls %4s avg %9.2f us  %5.1f%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
