#!/bin/bash
# round 5: one iteration of association-kernel work -- parity suites, A/B probes (OPA_ASSOC_SPEC x OPA_ASSOC_WAVES), phase timers
# usage: r5_iter.sh [tests|notests] ["spec waves" ...]
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_iter; mkdir -p $OUT
if [ "${1:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/tests.log
fi
shift
[ $# -eq 0 ] && set -- "1 12" "1 8"
{
for v in "$@"; do
  set -- $v
  echo "##### OPA_ASSOC_SPEC=$1 OPA_ASSOC_WAVES=$2"
  for cfg in "coco --alternate --check --trace 3" "wholebody --alternate --check"; do
    echo "=== r3_probe.py --config $cfg"; OPA_ASSOC_SPEC=$1 OPA_ASSOC_WAVES=$2 timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
  done
done
} > $OUT/probe.log 2>&1
grep -E "^#####|^===|cifcaf_assoc|^batch:|parity|level walk|Error|error|assert" $OUT/probe.log
if [ -f openpifpaf_amd/lib/libopenpifpaf_amd_ph.so ]; then bash tools/gpu/r3_phase.sh coco wholebody > $OUT/phase.log 2>&1; grep -E "^===|image 3|walk|batch:|pop\+entry" $OUT/phase.log | head -60; fi
