#!/bin/bash
# round 6, first call: where the round starts -- the decode of 32 and of 256 COCO images in ONE call_batch, per kernel
mkdir -p gpurun_out/r6
python tools/gpu/r3_probe.py --config coco --alternate --check > gpurun_out/r6/base_b32.log 2>&1
python tools/gpu/r3_probe.py --config coco --batch 256 --alternate > gpurun_out/r6/base_b256.log 2>&1
python tools/gpu/r3_probe.py --config coco --batch 128 --alternate > gpurun_out/r6/base_b128.log 2>&1
OPA_SEED_TIES=libstdcxx-fused python tools/gpu/r3_probe.py --config coco --batch 256 --alternate > gpurun_out/r6/base_b256_fused.log 2>&1
tail -n 30 gpurun_out/r6/base_b32.log gpurun_out/r6/base_b256.log | cut -c1-400
