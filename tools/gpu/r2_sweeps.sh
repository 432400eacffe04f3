#!/bin/bash
# round 2: randomised parity sweeps and the repeat stress of the final association kernel, with 8, 12 and 16 waves
# (counts per call: SWEEP_SCALE=1 is the full sweep of profiles/r2/parity_sweep.log's first run; the refresh after the
# last kernel changes ran at SWEEP_SCALE=0.4 to fit the GPU budget)
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1
S=${SWEEP_SCALE:-1}
n() { python -c "print(max(8, int($1 * $S)))"; }
{
for w in 12 8 16; do
  echo "== OPA_ASSOC_WAVES=$w"
  OPA_ASSOC_WAVES=$w timeout 900 python tools/gpu/parity_sweep.py $(n 250) $w coco 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 600 python tools/gpu/parity_sweep.py $(n 150) 21 dense 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/gpu/parity_sweep.py $(n 150) 22 tracking 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python tools/gpu/parity_sweep.py $(n 40) 23 wholebody 2>&1 | grep -v amdgpu.ids | tail -2
for w in 12 16; do echo "== repeat stress, OPA_ASSOC_WAVES=$w"; OPA_ASSOC_WAVES=$w timeout 600 python tools/gpu/stress_repeat.py $(n 200) 2>&1 | grep -v amdgpu.ids | tail -7; done
} > gpurun_out/r2_sweeps.log 2>&1
cat gpurun_out/r2_sweeps.log
