#!/bin/bash
# round 5 (second session), evidence on the FINAL sources after the prediction walk, the lookahead and the tighter collision test went in:
# GPU suite, smoke, rocprofv3 statistics + HBM PMC passes (collect_profiles stages 1-2), the driver's bench command, probes with
# traces, the session's three libraries side by side (first session's HEAD, the mid-session evidence run's sources, final).
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5b_final2; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
STAGES="1 2" ROUND=r5 bash tools/collect_profiles.sh > gpurun_out/collect_r5.log 2>&1; tail -3 gpurun_out/collect_r5.log
mkdir -p profiles/r5 && cp gpurun_out/prof_r5/pmc_traffic.json profiles/r5/pmc_traffic.json    # (bench.py reads it from there)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r5.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench_r5.json; cut -c1-1100 $OUT/bench_r5.json; cp bench_detail.json $OUT/bench_detail.json
{
for cfg in "coco --alternate --check --trace 3" "coco --fc --alternate --check" "wholebody --alternate --check --trace 3" "wholebody --fc --alternate --check"; do
  echo "=== r3_probe.py --config $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
done
} > $OUT/probe_all_workloads.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity" $OUT/probe_all_workloads.log
CONFIGS="coco --alternate --check;wholebody --alternate --check" bash tools/gpu/r5b_ab.sh "base|" "head|" "default|" "default|OPA_ASSOC_PREDICT=0 OPA_ASSOC_LOOKAHEAD=0 OPA_ASSOC_COLLIDE_SHIFT=0" "base|" "head|" "default|" > $OUT/assoc_before_after.txt 2>&1
cp gpurun_out/r5b_ab/probe.log $OUT/assoc_before_after.log; sed 's/cif_active.*cifcaf_assoc/assoc/' $OUT/assoc_before_after.txt
