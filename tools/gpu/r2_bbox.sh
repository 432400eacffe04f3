#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/bbox_tests.log 2>&1
echo "parity tests rc=$?" > gpurun_out/bbox_probe.log
{
echo "=== chunk boxes off"; OPA_ASSOC_BBOX=0 timeout 200 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -16
echo "=== chunk boxes on"; timeout 200 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -16
} >> gpurun_out/bbox_probe.log 2>&1
tail -n 5 gpurun_out/bbox_tests.log
cat gpurun_out/bbox_probe.log
