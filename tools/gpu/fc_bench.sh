#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for extra in "" "--force-complete"; do
timeout 900 python bench.py --decode-only --steps 10 --warmup 2 --cpu-seconds 8 $extra 2>>gpurun_out/bench.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$extra decode-only: value', d['value'], 'assoc ms', r['kernels']['cifcaf_assoc_kernel']['ms'], 'decode', r['decode_path']['ms_per_batch'], 'cafscored', r['kernels']['cafscored_kernel'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['all_cores_value'], 'ann', d['config']['annotations_per_batch'])"
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fields network 2>>gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('network fields: value', d['value'], 'ms/step', d['ms_per_step'])"
