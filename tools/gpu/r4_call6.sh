#!/bin/bash
# round 4, call 6: tie pass inside the association kernel (A/B against the separate launch), timers off by default
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call6; mkdir -p $OUT
run_probe() { echo "=== $1"; shift; timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids; }
{
run_probe "coco fused ties, timers off" OPA_ASSOC_TIMING=0 python tools/gpu/r3_probe.py --config coco --alternate --check
run_probe "coco separate tie launch, timers off" OPA_ASSOC_TIMING=0 OPA_FUSE_TIES=0 python tools/gpu/r3_probe.py --config coco --alternate --check
run_probe "coco fused ties, timers on" python tools/gpu/r3_probe.py --config coco --alternate
run_probe "wb fused ties" OPA_ASSOC_TIMING=0 python tools/gpu/r3_probe.py --config wholebody --alternate --check
run_probe "wb separate" OPA_ASSOC_TIMING=0 OPA_FUSE_TIES=0 python tools/gpu/r3_probe.py --config wholebody --alternate
} > $OUT/probe.log 2>&1
grep -E "^===|cifcaf_assoc|^wall|^batch:|parity|rror" $OUT/probe.log
timeout 1500 python -m pytest tests/test_gpu_ties.py tests/test_gpu_large_fields.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity.py tests/test_tracking_setup.py tests/test_torchscript_binding.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/tests.log
timeout 600 python bench.py --no-bf16-leg --steps 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; cat $OUT/bench.json; cp bench_detail.json $OUT/
