#!/bin/bash
# The 8-GPU form of BASELINE configs[4] (256 images = 32 per GPU, one rank per MI355X, RCCL over xGMI):
#     bash tools/launch_8gpu.sh [N] [extra bench.py flags]
# This is exactly what the driver runs for its scaling curve (N = 1, 2, 4, 8).
N=${1:-8}; shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0          # the host driver only supports dmabuf IPC
cd "$(dirname "$0")/.."
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29511}" \
    bench.py --gpus "$N" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" "$@"
