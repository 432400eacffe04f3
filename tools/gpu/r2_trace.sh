#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
OPA_TRACE_IMAGE=3 timeout 200 python tools/gpu/assoc_probe.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/trace3.log
cat gpurun_out/trace3.log | tail -45
