#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call10; mkdir -p $OUT
timeout 600 python tools/gpu/predictor_probe.py --batches 16 2>&1 | grep -v "amdgpu.ids\|UserWarning\|frame = " | tee $OUT/predictor_probe.log
