#!/bin/bash
# round 6, SECOND session (Winograd network kernels, task-slot padding), evidence on the final sources: the -m gpu suite, smoke, rocprofv3 statistics + PMC passes + FETCH_SIZE calibration
# (tools/collect_profiles.sh), the driver's bench command (after the PMC file is in place: its `traffic` is taken from a file measured
# on THESE kernel sources), probes, lanes sweep, randomised parity sweeps against the reference itself (oracle/_ref), repeat stress.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6b_final; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 400 python tools/gpu/winograd_probe.py --variants 0,2,3 --orders 0 2>&1 | grep -v amdgpu.ids > $OUT/winograd_probe.log; tail -4 $OUT/winograd_probe.log | cut -c1-330
ROUND=r6b STAGES="${STAGES:-1 2 3 4 5}" bash tools/collect_profiles.sh > gpurun_out/collect_r6b.log 2>&1; tail -3 gpurun_out/collect_r6b.log
mkdir -p profiles/r6 && cp gpurun_out/prof_r6b/pmc_traffic.json profiles/r6/pmc_traffic.json    # (bench.py reads it from there)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r6b.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r6b.json; cut -c1-1500 $OUT/bench_r6b.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --bench-batches --check --trace 3" "coco --batch 256 --alternate --bench-batches" "coco --batch 512 --alternate --bench-batches" "coco --fc --alternate --bench-batches --check" "wholebody --alternate --bench-batches --check --trace 3" "wholebody --fc --alternate --bench-batches --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity" $OUT/probe_all_workloads.log | cut -c1-300
TAG=../r6b_final bash tools/gpu/r6_lanes.sh > /dev/null 2>&1; cat $OUT/lanes.log
{
for SB in 261 361; do
  echo "== seeds $SB.."
  timeout 600 python tools/gpu/parity_sweep.py 200 $((SB + 0)) coco 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 1)) dense 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 2)) tracking 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 900 python tools/gpu/parity_sweep.py 50 $((SB + 3)) wholebody 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== repeat stress"; timeout 600 python tools/gpu/stress_repeat.py 150 2>&1 | grep -v amdgpu.ids | tail -4
} 2>&1 | tee $OUT/parity_sweep.log
