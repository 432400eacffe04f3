#!/bin/bash
# round 3: randomised parity sweeps (shapes, strides, crowd sizes, 11 option sets incl. the reference benchmark's
# force-complete setting) and the repeat stress of the final kernels; SWEEP_SCALE scales the counts
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1
S=${SWEEP_SCALE:-1}
n() { python -c "print(max(8, int($1 * $S)))"; }
{
timeout 900 python tools/gpu/parity_sweep.py $(n 300) 31 coco 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/gpu/parity_sweep.py $(n 150) 32 dense 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/gpu/parity_sweep.py $(n 150) 33 tracking 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python tools/gpu/parity_sweep.py $(n 50) 34 wholebody 2>&1 | grep -v amdgpu.ids | tail -2
echo "== repeat stress"; timeout 600 python tools/gpu/stress_repeat.py $(n 200) 2>&1 | grep -v amdgpu.ids | tail -7
} > gpurun_out/r3_sweeps.log 2>&1
cat gpurun_out/r3_sweeps.log
