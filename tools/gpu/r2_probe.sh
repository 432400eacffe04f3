#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "=== probe (default waves)"; OPA_TRACE_IMAGE=3 timeout 300 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids
for w in 8 12; do echo "=== probe waves=$w"; OPA_ASSOC_WAVES=$w timeout 120 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -12; done
} > gpurun_out/call4_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity_r2.py -q -m gpu -s > gpurun_out/call4_r2tests.log 2>&1
echo "r2 tests rc=$?" >> gpurun_out/call4_probe.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/call4_gputests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/call4_probe.log
tail -n 8 gpurun_out/call4_r2tests.log; tail -n 8 gpurun_out/call4_gputests.log
cat gpurun_out/call4_probe.log
