#!/bin/bash
# round 5: GPU_MAX_HW_QUEUES x decode lanes (bench.py --decode-only --decode-streams N), and the depth-limit tie test
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_queues; mkdir -p $OUT
timeout 600 python -m pytest tests/test_tie_depth_limit.py tests/test_gpu_ties.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/tests.log
{
for q in 4 8 16; do
  for n in 2 4 8 12; do
    v=$(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --decode-only --decode-streams $n --steps 96 --warmup 52 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])")
    echo "GPU_MAX_HW_QUEUES=$q lanes $n images/s $v"
  done
done
} 2>&1 | tee $OUT/hw_queues.log
