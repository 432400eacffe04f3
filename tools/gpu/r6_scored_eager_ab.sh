#!/bin/bash
# round 6, second session: A/B of a CafScored::fill variant built as lib/libopenpifpaf_amd_eager.so (build.build_diagnostic(<define>, 'eager',
# source='cafscored.hip')) against the default library: -DOPA_SCORED_EAGER=1 (the seven plane loads travel together; the optimiser sinks
# six of them behind the threshold test otherwise) and -DOPA_SCORED_PREFETCH=1 / 0 (the next step's confidence one step ahead);
# 32 / 256 images, the force-complete pass, wholebody; parity checked.
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
