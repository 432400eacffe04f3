#!/bin/bash
# round 5 (second session): A/B of association-kernel variants.  Each argument: "<library suffix or 'default'>|<env assignments>",
# e.g. r5b_ab.sh "base|" "default|" "default|OPA_ASSOC_PREDEDUP=0" "p4|".  Libraries: openpifpaf_amd/lib/libopenpifpaf_amd_<suffix>.so
# CONFIGS (env): the probe configurations, ';'-separated (default: coco with the crowded image's trace; wholebody)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5b_ab; mkdir -p $OUT
CONFIGS="${CONFIGS:-coco --alternate --check --trace 3;wholebody --alternate --check}"
{
for v in "$@"; do
  lib="${v%%|*}"; envs="${v#*|}"
  if [ "$lib" = default ]; then L=""; else L="OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$lib.so"; fi
  echo "##### $lib | $envs"
  IFS=';' read -ra CF <<< "$CONFIGS"
  for cfg in "${CF[@]}"; do
    echo "=== r3_probe.py --config $cfg"; env $L $envs timeout 300 python tools/gpu/r3_probe.py --config $cfg 2>&1 | grep -v amdgpu.ids
  done
done
} > $OUT/probe.log 2>&1
grep -E "^#####|^===|cifcaf_assoc|^batch:|parity|Error|error|assert|^ +(3|19|27) +[0-9]+ +[0-9]+ \|" $OUT/probe.log
