#!/bin/bash
# round 5: A/B of association-kernel switches given as env assignments, e.g. r5_ab.sh "OPA_ASSOC_EARLY=0" "OPA_ASSOC_EARLY=1 OPA_ASSOC_COMMIT_RUN=8"
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_ab; mkdir -p $OUT
{
for v in "$@"; do
  echo "##### $v"
  for cfg in "coco --alternate --check --trace 3" "wholebody --alternate --check"; do
    echo "=== r3_probe.py --config $cfg"; env $v timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
  done
done
} > $OUT/probe.log 2>&1
grep -E "^#####|^===|cifcaf_assoc|^batch:|parity|Error|error|assert" $OUT/probe.log
