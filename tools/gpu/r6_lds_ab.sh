#!/bin/bash
# round 6, review item 7 (LDS bank conflicts of the association kernel): the task slots at 72 bytes (default) against 64
# (-DOPA_TASK_SLOT_PAD=0) and other distances between the growers' private blocks (-DOPA_ASSOC_PRIVATE_PAD=16 / 144), three
# probe runs each on the bench's batches (association kernel time by HIP events, parity against the reference in every run).
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6/lds_ab; mkdir -p $OUT
export PYTHONPATH=. PYTHONUNBUFFERED=1
for lib in default slot64 pad16 pad144; do
  if [ $lib = default ]; then unset OPA_LIB_PATH; else export OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$lib.so; fi
  for rep in 1 2 3; do
    for cfg in "coco --alternate --bench-batches --check" "wholebody --alternate --bench-batches --check"; do
      echo "=== $lib run $rep: $cfg"
      timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -E "decode|parity|Error|error" | cut -c1-300
    done
  done
done 2>&1 | tee $OUT/lds_ab.log
