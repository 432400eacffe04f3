#!/bin/bash
# one iteration of association-kernel work: parity + stress + phase timers + decode-only bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/gpu/stress_repeat.py ${1:-40} 2>&1 | tail -7
OPA_LIB_PATH=openpifpaf_amd/lib/libopa_timing.so timeout 120 python tools/assoc_timing.py 2>&1 | sed 's/.*| us:/us:/' | tail -8
timeout 300 python bench.py --no-cpu-baseline --decode-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('decode-only', d['value'], 'img/s', d['ms_per_step'], 'ms; assoc', d['roofline']['kernels']['cifcaf_assoc_kernel'])"
