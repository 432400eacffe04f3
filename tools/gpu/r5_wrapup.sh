#!/bin/bash
# round 5, wrap-up after the last source change: GPU suite, rocprofv3 + PMC collection (stages 1-4), the driver's bench command
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r5_final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/gpu_tests.log
ROUND=r5 bash tools/collect_profiles.sh > gpurun_out/collect_r5.log 2>&1; tail -3 gpurun_out/collect_r5.log
mkdir -p profiles/r5 && cp gpurun_out/prof_r5/pmc_traffic.json profiles/r5/pmc_traffic.json    # (bench.py reads it from there)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r5.json 2> $OUT/bench.err; echo "bench rc $?"; cut -c1-900 $OUT/bench_r5.json; cp bench_detail.json $OUT/bench_detail.json
