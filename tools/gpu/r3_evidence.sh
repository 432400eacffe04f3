#!/bin/bash
# Round-3 evidence that is not rocprofv3 output: per-workload probes with the parity check, commit traces of the
# slowest images, the growers' phase timers (diagnostic library), and the default bench line.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3_evidence; rm -rf $OUT; mkdir -p $OUT
for cfg in "coco --alternate" "coco --fc --alternate" "wholebody --alternate" "wholebody --fc --alternate"; do
  echo "=== r3_probe.py --config $cfg --check"; timeout 400 python tools/gpu/r3_probe.py --config $cfg --check 2>&1 | grep -v amdgpu.ids
done > $OUT/probe_all_workloads.log 2>&1
{ echo "# python tools/gpu/r3_probe.py --config wholebody --trace 3"; timeout 300 python tools/gpu/r3_probe.py --config wholebody --trace 3 2>&1 | grep -v amdgpu.ids; } > $OUT/wholebody_trace.log
{ echo "# python tools/gpu/r3_probe.py --config coco --trace 3"; timeout 300 python tools/gpu/r3_probe.py --config coco --trace 3 2>&1 | grep -v amdgpu.ids; } > $OUT/coco_trace.log
if [ -f openpifpaf_amd/lib/libopenpifpaf_amd_ph.so ]; then
  bash tools/gpu/r3_phase.sh wholebody coco > $OUT/assoc_phase_timers.log 2>&1
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r3.json 2> $OUT/bench_r3.err
tail -2 $OUT/bench_r3.err; wc -c $OUT/*
