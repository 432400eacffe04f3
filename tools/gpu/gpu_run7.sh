#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
# N>1 control flow on one GPU (gloo, both ranks on cuda:0): validates sharding/gather/timing logic of bench.py
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --dist-backend gloo --share-device --no-cpu-baseline 2>gpurun_out/bench2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2-rank gloo: n_gpus', d['n_gpus'], 'value', d['value'], 'global_batch', d['config']['global_batch'])"
tail -3 gpurun_out/bench2.err
