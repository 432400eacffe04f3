#!/bin/bash
# round 6: rocprofv3 kernel statistics of the decode of ${BATCH:-256} (and 32) COCO images in one call_batch
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"; OUT="$REPO/gpurun_out/r6/${TAG:-prof}"; mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH="$REPO"
cd /tmp
for B in ${BATCHES:-256 32}; do
  rm -rf /tmp/prof_b$B
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$B -o stats -- \
      python "$REPO/tools/gpu/r3_probe.py" --config coco --batch $B --alternate --reps 12 ${PROBE_ARGS} > "$OUT/b${B}_stdout.log" 2> "$OUT/b${B}_stderr.log"
  f=$(find /tmp/prof_b$B -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/b${B}_kernel_stats.csv"
  echo "== batch $B"; python - "$OUT/b${B}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r['Name'].split('(')[0].replace('opa::', '').replace('void ', '')[:60]
    if 'at::' in name or 'elementwise' in name: continue
    print('%-60s calls %4s avg %9.1f us  min %9.1f  max %9.1f' % (name, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
