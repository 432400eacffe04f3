#!/bin/bash
# round 4, call 9: lanes with the tie pass fused, hardware queues, full suite after the RAW revert
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call9; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.log
timeout 900 python bench.py --no-bf16-leg --steps 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; cat $OUT/bench.json | cut -c1-3000; cp bench_detail.json $OUT/; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open('bench_detail.json'))
print('lanes', json.dumps(d['configs']['decode_two_in_flight']['decode_only_images_per_s']))
print('predictor', {k: v for k, v in d['configs']['predictor'].items() if k != 'what'})
PY
for q in 8 16; do
  echo "=== GPU_MAX_HW_QUEUES=$q"
  for n in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --decode-only --decode-streams $n --steps 96 --warmup 40 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lanes', $n, 'images/s', d['value'])"
  done
done 2>&1 | tee $OUT/hw_queues.log
