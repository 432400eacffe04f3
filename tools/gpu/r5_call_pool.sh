#!/bin/bash
# round 5: tile pool with a spill region + automatic sizing -- GPU suite, the wholebody / coco sweeps (launches repeated with a full pool are counted), smoke
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_pool; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/tests.log
{
timeout 600 python tools/gpu/parity_sweep.py 75 34 wholebody 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/gpu/parity_sweep.py 60 134 wholebody 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/gpu/parity_sweep.py 150 31 coco 2>&1 | grep -v amdgpu.ids | tail -3
} 2>&1 | tee $OUT/sweeps.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
