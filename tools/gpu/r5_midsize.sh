#!/bin/bash
# round 5: mid-size kernels -- stage parity tests, then per-kernel times of the default library and of diagnostic variants
# usage: r5_midsize.sh [suffix ...]   (libopenpifpaf_amd_<suffix>.so built by build.build_diagnostic)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_midsize; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r3.py tests/test_gpu_large_fields.py tests/test_gpu_ties.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/tests.log
{
for v in default "$@"; do
  echo "##### library: $v"
  for cfg in "coco --alternate --check" "coco --fc --alternate --check" "wholebody --alternate --check"; do
    if [ "$v" = default ]; then L=""; else L="OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$v.so"; fi
    echo "=== $cfg"; env $L timeout 300 python tools/gpu/r3_probe.py --config $cfg --reps 20 2>&1 | grep -E "cif_active|parity|Error|error"
  done
done
} 2>&1 | tee $OUT/probe.log
