#!/bin/bash
# round 4, call 8: tile pool without the fence in cif_active, force complete on the CAF field itself (RAW lists)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call8; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/tests.log
{
echo "=== coco"; timeout 300 python tools/gpu/r3_probe.py --config coco --alternate --check 2>&1 | grep -v amdgpu.ids
echo "=== coco fc"; timeout 300 python tools/gpu/r3_probe.py --config coco --fc --alternate --check 2>&1 | grep -v amdgpu.ids
echo "=== wb"; timeout 300 python tools/gpu/r3_probe.py --config wholebody --alternate --check 2>&1 | grep -v amdgpu.ids
echo "=== wb fc"; timeout 300 python tools/gpu/r3_probe.py --config wholebody --fc --alternate --check 2>&1 | grep -v amdgpu.ids
} > $OUT/probe.log 2>&1
grep -E "^===|config |cifcaf_assoc|^wall|^batch:|parity|rror" $OUT/probe.log
timeout 900 python bench.py --no-bf16-leg --steps 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; cat $OUT/bench.json; cp bench_detail.json $OUT/; tail -3 $OUT/bench.err
