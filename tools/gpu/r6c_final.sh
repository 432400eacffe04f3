#!/bin/bash
# round 6, THIRD session (float32 1x1 convolutions on the bf16 MFMA pipe with split operands), evidence on the final sources: the
# -m gpu suite, smoke, the GEMM probe, rocprofv3 statistics + PMC passes + step traces (tools/collect_profiles.sh), the driver's bench
# command (after the PMC file is in place), probes of every workload, one set of randomised parity sweeps against oracle/_ref.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6c_final; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 300 python tools/gpu/gemm_x3_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_x3_probe.log; tail -9 $OUT/gemm_x3_probe.log | cut -c1-250
ROUND=r6c STAGES="${STAGES:-1 2 3}" bash tools/collect_profiles.sh > gpurun_out/collect_r6c.log 2>&1; tail -3 gpurun_out/collect_r6c.log
mkdir -p profiles/r6 && cp gpurun_out/prof_r6c/pmc_traffic.json profiles/r6/pmc_traffic.json    # (bench.py reads it from there)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r6c.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r6c.json; cut -c1-1800 $OUT/bench_r6c.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --bench-batches --check" "coco --batch 256 --alternate --bench-batches" "coco --fc --alternate --bench-batches --check" "wholebody --alternate --bench-batches --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v "amdgpu.ids\|^status"
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|parity" $OUT/probe_all_workloads.log | cut -c1-300
{
SB=461
timeout 600 python tools/gpu/parity_sweep.py 200 $((SB + 0)) coco 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 1)) dense 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 2)) tracking 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python tools/gpu/parity_sweep.py 50 $((SB + 3)) wholebody 2>&1 | grep -v amdgpu.ids | tail -1
} 2>&1 | tee $OUT/parity_sweep.log
