#!/bin/bash
# round 5 (second session), final evidence on the final sources: the -m gpu suite, smoke, rocprofv3 statistics + PMC passes
# (tools/collect_profiles.sh), the driver's bench command (after the PMC file is in place: its `traffic` is taken from a file
# measured on THESE kernel sources), the probes with commit traces, the round's start (lib/libopenpifpaf_amd_base.so = HEAD of
# the first session) against the result, randomised parity sweeps (two seed sets) and the repeat stress.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5b_final; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
ROUND=r5 bash tools/collect_profiles.sh > gpurun_out/collect_r5.log 2>&1; tail -3 gpurun_out/collect_r5.log
mkdir -p profiles/r5 && cp gpurun_out/prof_r5/pmc_traffic.json profiles/r5/pmc_traffic.json    # (bench.py reads it from there)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r5.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r5.json; cut -c1-1200 $OUT/bench_r5.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --check --trace 3" "coco --fc --alternate --check" "wholebody --alternate --check --trace 3" "wholebody --fc --alternate --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity" $OUT/probe_all_workloads.log
if [ -f openpifpaf_amd/lib/libopenpifpaf_amd_base.so ]; then
  CONFIGS="coco --alternate --check;wholebody --alternate --check" bash tools/gpu/r5b_ab.sh "base|" "default|" "default|OPA_ASSOC_PREDEDUP=0" "base|" "default|" > $OUT/assoc_before_after.txt 2>&1
  cp gpurun_out/r5b_ab/probe.log $OUT/assoc_before_after.log; cat $OUT/assoc_before_after.txt | sed 's/cif_active.*cifcaf_assoc/assoc/'
fi
{
for SB in 41 141; do
  echo "== seeds $SB.."
  timeout 600 python tools/gpu/parity_sweep.py 200 $((SB + 0)) coco 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 1)) dense 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python tools/gpu/parity_sweep.py 100 $((SB + 2)) tracking 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 900 python tools/gpu/parity_sweep.py 50 $((SB + 3)) wholebody 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== repeat stress"; timeout 600 python tools/gpu/stress_repeat.py 150 2>&1 | grep -v amdgpu.ids | tail -4
} 2>&1 | tee $OUT/parity_sweep.log
