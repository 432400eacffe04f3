#!/bin/bash
# one iteration of association-kernel work: parity tests, then the probe (statistics + per-kernel times)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/iter_tests.log 2>&1
echo "parity tests rc=$?" > gpurun_out/iter_probe.log
OPA_TRACE_IMAGE=3 timeout 200 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/iter_probe.log 2>&1
tail -n 4 gpurun_out/iter_tests.log
head -n 18 gpurun_out/iter_probe.log
