#!/bin/bash
# round 6: cafscored variants (cells per thread and step, prefetch) x what the step does (all, no map gathers, nothing kept)
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONPATH=. PYTHONUNBUFFERED=1
for lib in ${LIBS:-default sc1 sc1np sc2 sc2np}; do
  for par in "" "--param caf_threshold=0.99"; do
    if [ $lib = default ]; then unset OPA_LIB_PATH; else export OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$lib.so; fi
    echo "lib $lib $par: $(timeout 300 python tools/gpu/r3_probe.py --config coco --batch 256 --alternate $par 2>&1 | grep -o 'cafscored [0-9.]* us')"
  done
done 2>&1 | tee gpurun_out/r6/scored_ab.log
