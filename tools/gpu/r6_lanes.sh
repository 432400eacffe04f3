#!/bin/bash
# round 6: decode-only throughput over (images per call_batch) x (decode lanes)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6/${TAG:-lanes}; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
for B in ${BATCHES:-32 64 128 256}; do
  for n in ${LANES:-1 2 3 4}; do
    steps=$(( 6144 / B )); [ $steps -lt 24 ] && steps=24
    v=$(timeout 300 python bench.py --decode-only --decode-streams $n --batch $B --steps $steps --warmup $(( 2 * n + 4 )) --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "batch $B lanes $n images/s $v"
  done
done 2>&1 | tee $OUT/lanes.log
