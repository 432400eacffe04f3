#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
timeout 300 python bench.py --steps 10 --warmup 2 --decode-only --no-cpu-baseline 2>>gpurun_out/bench.err | tee gpurun_out/bench_decode_only.json
