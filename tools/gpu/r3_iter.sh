#!/bin/bash
# One round-3 iteration on the GPU box: the parity suites (or the whole -m gpu suite with FULL=1), then the probes
# (PROBES="coco;coco --fc;wholebody;wholebody --fc" by default).
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3_iter; mkdir -p $OUT
if [ "${FULL:-0}" = "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.log
else
  timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.log
fi
IFS=';' read -ra LIST <<< "${PROBES:-coco;coco --fc;wholebody;wholebody --fc}"
for cfg in "${LIST[@]}"; do
  echo "=== $cfg"; timeout 300 python tools/gpu/r3_probe.py --config $cfg --check 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe.log
done
