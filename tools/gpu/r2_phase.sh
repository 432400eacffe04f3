#!/bin/bash
# phase timing of the growers (diagnostic library built with -DOPA_ASSOC_PHASE_TIMING)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "=== production"; timeout 200 python tools/gpu/assoc_probe.py 8 2>&1 | grep -v amdgpu.ids | tail -4
echo "=== phase timing"; OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_ph.so timeout 300 python tools/gpu/assoc_probe.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/phase_raw.log
grep -v PHASE gpurun_out/phase_raw.log | head -14
# the last launch's phase lines (the probe launches many times): keep the final 40
grep PHASE gpurun_out/phase_raw.log | tail -40
} > gpurun_out/phase.log 2>&1
cat gpurun_out/phase.log
