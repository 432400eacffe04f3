#!/bin/bash
# round 5, final evidence: the -m gpu suite, smoke, the driver's bench command, the probes with commit traces, the
# randomised parity sweeps (two seed sets) and the repeat stress
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_final; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r5.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r5.json; cat $OUT/bench_r5.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --check --trace 3" "coco --fc --alternate --check" "wholebody --alternate --check --trace 3" "wholebody --fc --alternate --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity" $OUT/probe_all_workloads.log
{
for SB in 31 131; do
  echo "== seeds $SB.."
  timeout 600 python tools/gpu/parity_sweep.py 200 $((SB + 0)) coco 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 1)) dense 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 2)) tracking 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 900 python tools/gpu/parity_sweep.py 50 $((SB + 3)) wholebody 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== repeat stress"; timeout 600 python tools/gpu/stress_repeat.py 150 2>&1 | grep -v amdgpu.ids | tail -4
} 2>&1 | tee $OUT/parity_sweep.log
timeout 300 python tools/gpu/gemm_f32_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_f32_probe.log
