#!/bin/bash
# End-of-round evidence: GPU tests, smoke, default bench line, rocprofv3 summaries.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.log
( time timeout 900 python bench.py 2>gpurun_out/bench.err ) 2>gpurun_out/bench.time | tee gpurun_out/bench.json
grep -E "real|warm-up" gpurun_out/bench.time gpurun_out/bench.err
[ -f openpifpaf_amd/lib/libopa_timing.so ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
    -DOPA_ASSOC_TIMING -o openpifpaf_amd/lib/libopa_timing.so openpifpaf_amd/csrc/*.hip 2>/dev/null
OPA_LIB_PATH=openpifpaf_amd/lib/libopa_timing.so timeout 120 python tools/assoc_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/assoc_timing.log
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
tail -3 gpurun_out/collect.log
