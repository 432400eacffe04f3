#!/bin/bash
# round 5, call 1: the level walk (batched bone evaluation) -- parity suite, then A/B against every bone on demand, 12 and 8 waves
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_call1; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.log
{
for v in "1 12" "1 8" "0 12"; do
  set -- $v
  echo "##### OPA_ASSOC_SPEC=$1 OPA_ASSOC_WAVES=$2"
  for cfg in "coco --alternate --check --trace 3" "wholebody --alternate --check"; do
    echo "=== r3_probe.py --config $cfg"; OPA_ASSOC_SPEC=$1 OPA_ASSOC_WAVES=$2 timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
  done
done
} > $OUT/probe.log 2>&1
grep -E "^#####|^===|cifcaf_assoc|^wall|^batch:|parity|level walk|Error|error|assert" $OUT/probe.log
