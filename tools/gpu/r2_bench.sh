#!/bin/bash
# round 2: the bench lines of every BASELINE configuration (N=1) + the two-process test
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > gpurun_out/bench_config2.json 2> gpurun_out/bench_config2.err; echo "config2 rc=$?"
timeout 900 python bench.py --config 3 --steps 8 --warmup 2 > gpurun_out/bench_config3.json 2> gpurun_out/bench_config3.err; echo "config3 rc=$?"
timeout 900 python bench.py --config 4 --steps 8 --warmup 2 > gpurun_out/bench_config4.json 2> gpurun_out/bench_config4.err; echo "config4 rc=$?"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5
for f in gpurun_out/bench_config*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
except Exception as e:
    print('no json', e); sys.exit(0)
print('value', d['value'], 'ms/step', d['ms_per_step'], 'vs_baseline', d['vs_baseline'], 'bf16', d.get('bf16_backbone'))
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms')}, d['roofline']['decode_path'], d['roofline'].get('backbone_mfma'))
print('cpu', d['cpu_baseline'] and {k: d['cpu_baseline'][k] for k in ('value', 'fresh_instance_value', 'all_cores_value', 'all_cores')})
print('ref pipeline', d['reference_pipeline'])
PY
done
tail -3 gpurun_out/bench_config*.err
