#!/bin/bash
# after a kernel change: parity first; if green, the rocprofv3 evidence, then the headline bench line stamped with it
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_host_api.py -q -m gpu -x > gpurun_out/final_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/final_tests.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED: nothing else run"; exit 1; fi
bash tools/collect_profiles.sh > /dev/null 2>&1
cp gpurun_out/prof_r2/pmc_traffic.json profiles/r2/pmc_traffic.json
grep -A9 "decode-only --steps 10" gpurun_out/prof_r2/summary.md | tail -8
timeout 500 python bench.py > gpurun_out/bench_config2.json 2> gpurun_out/bench_config2.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/bench_config2.json") if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "vs_baseline", d["vs_baseline"], "traffic", d["roofline"]["traffic"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["decode_path"])
print(d["bf16_backbone"])
PY
