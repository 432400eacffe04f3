#!/bin/bash
# first GPU contact: build check, parity tests (no -x: see every failure)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Marketing" | head -4 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
