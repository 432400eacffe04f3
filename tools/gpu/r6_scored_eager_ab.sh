#!/bin/bash
# round 6, second session: CafScored::fill with its seven plane loads travelling together (-DOPA_SCORED_EAGER=1; the optimiser sinks six
# of them behind the threshold test otherwise) against the default, 32 / 256 images and the force-complete pass, parity checked.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6/scored_eager; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
for lib in default eager default eager; do
  if [ $lib = default ]; then unset OPA_LIB_PATH; else export OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$lib.so; fi
  for cfg in "coco --alternate --bench-batches --check" "coco --batch 256 --alternate --bench-batches" "coco --fc --alternate --bench-batches --check" "wholebody --alternate --bench-batches --check"; do
    echo "=== $lib: $cfg"
    timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -E "decode|parity|Error|error" | cut -c1-300
  done
done 2>&1 | tee $OUT/ab.log
