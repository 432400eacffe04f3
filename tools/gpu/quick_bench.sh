#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_epilogue.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'])"
grep warm-up gpurun_out/bench.err
