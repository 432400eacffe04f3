#!/bin/bash
# round 4, call 4: 16-slot pool of the LDS variant (wholebody), tie pass with LDS sub-sorts, the new tests
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call4; mkdir -p $OUT
run_probe() { echo "=== $1"; shift; timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids; }
{
run_probe "wb pool16 subsort=1" python tools/gpu/r3_probe.py --config wholebody --alternate --check --trace 11
run_probe "wb pool16 subsort=0" OPA_TIE_SUBSORT=0 python tools/gpu/r3_probe.py --config wholebody --alternate
run_probe "wb fc" python tools/gpu/r3_probe.py --config wholebody --fc --alternate --check
run_probe "coco" python tools/gpu/r3_probe.py --config coco --alternate --check
} > $OUT/probe.log 2>&1
grep -E "^===|cifcaf_assoc|^batch:|parity|rror" $OUT/probe.log
timeout 1500 python -m pytest tests/test_gpu_ties.py tests/test_gpu_large_fields.py tests/test_gpu_host_api.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests.log
