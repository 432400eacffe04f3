#!/bin/bash
# round 4, call 3: collision stops + inherited predictions; the new host-side and large-field tests
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call3; mkdir -p $OUT
run_probe() { echo "=== $1"; shift; timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids; }
{
run_probe "coco collide=1 inherit=1" python tools/gpu/r3_probe.py --config coco --alternate --check --trace 3
run_probe "coco collide=1 inherit=0" OPA_ASSOC_INHERIT=0 python tools/gpu/r3_probe.py --config coco --alternate --check
run_probe "coco collide=0 inherit=1" OPA_ASSOC_COLLIDE=0 python tools/gpu/r3_probe.py --config coco --alternate
run_probe "wb collide=1 inherit=1" python tools/gpu/r3_probe.py --config wholebody --alternate --check --trace 11
run_probe "wb collide=1 inherit=0" OPA_ASSOC_INHERIT=0 python tools/gpu/r3_probe.py --config wholebody --alternate --check
run_probe "coco fc" python tools/gpu/r3_probe.py --config coco --fc --alternate --check
} > $OUT/probe.log 2>&1
grep -E "^===|cifcaf_assoc|^batch:|parity|rror" $OUT/probe.log
timeout 1200 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests_assoc.log
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_large_fields.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests_new.log
