#!/bin/bash
# tuning variants of csrc/gemm_f32x3.hip (each a second library, loaded through OPA_LIB_PATH) on the compute-bound shapes
cd "$(dirname "$0")/../.."
export SHAPES=${SHAPES:-layer3,layer4}
echo "== default"; python tools/gpu/gemm_x3_probe.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== $v"
  OPA_LIB_PATH=openpifpaf_amd/lib/libopenpifpaf_amd_$v.so python tools/gpu/gemm_x3_probe.py 2>&1 | grep -v amdgpu.ids
done
