#!/bin/bash
# round 4, first GPU call: state of HEAD -- parity suites, the probes with the coordinator breakdown, the default bench
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r2.py tests/test_gpu_ties.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.log
for cfg in "coco --alternate --trace 3" "wholebody --alternate --trace 11"; do
  echo "=== $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe.log
done
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench.json; cp bench_detail.json $OUT/ 2>/dev/null
tail -c 600 $OUT/bench.json
