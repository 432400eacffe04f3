#!/bin/bash
# N-GPU bench runs (one rank per MI355X, RCCL over xGMI; images shard one batch per GPU).  Default: the headline, config 2
# (resnet50, 32 images per GPU).  BASELINE configs[4] (shufflenetv2k16, 256 images = 32 per GPU on 8 GPUs) is
#     bash tools/launch_8gpu.sh 8 --config 3
#     bash tools/launch_8gpu.sh [N] [extra bench.py flags]
# This is exactly what the driver runs for its scaling curve (N = 1, 2, 4, 8).
N=${1:-8}; shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0          # the host driver only supports dmabuf IPC
cd "$(dirname "$0")/.."
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29511}" \
    bench.py --gpus "$N" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" "$@"
