#!/bin/bash
# full GPU test suite + decode-only bench + probe
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/full_tests.log 2>&1
echo "gpu tests rc=$?" > gpurun_out/full_summary.log
tail -n 6 gpurun_out/full_tests.log >> gpurun_out/full_summary.log
timeout 600 python bench.py --decode-only --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/full_summary.log
OPA_DECODE_SIDE_STREAM=0 timeout 600 python bench.py --decode-only --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/full_summary.log
timeout 200 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 3 >> gpurun_out/full_summary.log
cat gpurun_out/full_summary.log
