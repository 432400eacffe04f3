#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap prio: value', d['value'], 'ms/step', d['ms_per_step'])"
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap 2>>gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no overlap : value', d['value'], 'ms/step', d['ms_per_step'])"
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 64 2>>gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 64   : value', d['value'], 'ms/step', d['ms_per_step'])"
grep warm-up gpurun_out/bench.err
