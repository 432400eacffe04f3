#!/bin/bash
# round 4, final evidence: the -m gpu suite, smoke, the driver's bench command, the probes with commit traces
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_final; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r4.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r4.json; cat $OUT/bench_r4.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --check --trace 3" "coco --fc --alternate --check" "wholebody --alternate --check --trace 11" "wholebody --fc --alternate --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity" $OUT/probe_all_workloads.log
timeout 300 python tools/gpu/predictor_probe.py --batches 24 2>&1 | grep -v "amdgpu.ids\|UserWarning\|frame = " | tee $OUT/predictor_probe.log
