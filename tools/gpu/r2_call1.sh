#!/bin/bash
# round 2, GPU call 1: the new association kernel -- statistics, parity suites, A/B against the round-1 library
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "=== probe (default waves)"; timeout 300 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids
for w in 8 12; do echo "=== probe waves=$w"; OPA_ASSOC_WAVES=$w timeout 120 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -4; done
echo "=== r1 library, decode only"; OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_r1.so timeout 200 python bench.py --decode-only --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
echo "=== new library, decode only"; timeout 200 python bench.py --decode-only --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
} > gpurun_out/call1_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity_r2.py -x -q -m gpu -s > gpurun_out/call1_r2tests.log 2>&1
echo "r2 tests rc=$?" >> gpurun_out/call1_probe.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/call1_gputests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/call1_probe.log
tail -5 gpurun_out/call1_r2tests.log gpurun_out/call1_gputests.log
cat gpurun_out/call1_probe.log
