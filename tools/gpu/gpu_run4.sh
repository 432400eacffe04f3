#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
OPA_LIB_PATH=openpifpaf_amd/lib/libopa_timing.so timeout 120 python tools/assoc_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/assoc_timing.log
timeout 300 python bench.py --steps 10 --warmup 2 --decode-only --no-cpu-baseline 2>>gpurun_out/bench.err | tee gpurun_out/bench_decode_only.json
