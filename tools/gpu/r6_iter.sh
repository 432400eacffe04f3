#!/bin/bash
# round 6: one iteration -- the -m gpu suite, then the decode of 32 / 256 COCO images per kernel (r3_probe.py), optionally A/B
# against round 5's stage kernels (OPA_STAGE_WORKLIST=0, read once at library load)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6/${TAG:-iter}; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
if [ "${TESTS:-1}" = "1" ]; then timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -${TAIL:-15} | tee $OUT/gpu_tests.log; fi
for cfg in "coco --alternate --check" "coco --batch 256 --alternate" ${EXTRA_CFG:+"$EXTRA_CFG"}; do
  echo "=== r3_probe.py --config $cfg"
  timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -E "^config|decode|^wall|^batch:|parity|Error|error" | cut -c1-400
  if [ "${AB:-0}" = "1" ]; then
    echo "--- OPA_STAGE_WORKLIST=0"
    OPA_STAGE_WORKLIST=0 timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -E "decode|^wall|parity|Error|error" | cut -c1-400
  fi
done 2>&1 | tee $OUT/probe.log
