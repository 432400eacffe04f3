#!/bin/bash
# round 3: randomised parity sweeps (shapes, strides, crowd sizes, 11 option sets incl. the reference benchmark's
# force-complete setting) and the repeat stress of the final kernels; SWEEP_SCALE scales the counts
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1
S=${SWEEP_SCALE:-1}
SB=${SEED_BASE:-31}                # seeds SB .. SB+3 for the four sweeps (round 4 added a second set: 131)
n() { python -c "print(max(8, int($1 * $S)))"; }
{
timeout 900 python tools/gpu/parity_sweep.py $(n 300) $((SB + 0)) coco 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/gpu/parity_sweep.py $(n 150) $((SB + 1)) dense 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/gpu/parity_sweep.py $(n 150) $((SB + 2)) tracking 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python tools/gpu/parity_sweep.py $(n 50) $((SB + 3)) wholebody 2>&1 | grep -v amdgpu.ids | tail -2
echo "== repeat stress"; timeout 600 python tools/gpu/stress_repeat.py $(n 200) 2>&1 | grep -v amdgpu.ids | tail -7
} > gpurun_out/r3_sweeps.log 2>&1
cat gpurun_out/r3_sweeps.log
