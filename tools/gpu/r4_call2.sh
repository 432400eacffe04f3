#!/bin/bash
# round 4, call 2: lean coordinator (+ inherited predictions) against round 3's library; new host-side tests
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call2; mkdir -p $OUT
R3=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_r3.so
run_probe() { # name env... -- args
  echo "=== $1"; shift
  timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids
}
{
run_probe "coco r3"            OPA_LIB_PATH=$R3 python tools/gpu/r3_probe.py --config coco --alternate
run_probe "coco new inherit=0" OPA_ASSOC_INHERIT=0 python tools/gpu/r3_probe.py --config coco --alternate --check
run_probe "coco new inherit=1" OPA_ASSOC_INHERIT=1 python tools/gpu/r3_probe.py --config coco --alternate --check --trace 3
run_probe "wb r3"              OPA_LIB_PATH=$R3 python tools/gpu/r3_probe.py --config wholebody --alternate
run_probe "wb new inherit=0"   OPA_ASSOC_INHERIT=0 python tools/gpu/r3_probe.py --config wholebody --alternate --check
run_probe "wb new inherit=1"   OPA_ASSOC_INHERIT=1 python tools/gpu/r3_probe.py --config wholebody --alternate --check --trace 11
} > $OUT/probe.log 2>&1
grep -E "^===|cifcaf_assoc|^batch:|parity|Error|error" $OUT/probe.log
timeout 1200 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests_assoc.log
OPA_ASSOC_INHERIT=0 timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/tests_assoc_noinherit.log
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_large_fields.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests_new.log
