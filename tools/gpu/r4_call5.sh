#!/bin/bash
# round 4, call 5: the whole -m gpu suite and the default bench run (new legs: all_active, predictor, batch-1 breakdown)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4_call5; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench.json; cp bench_detail.json $OUT/ 2>/dev/null
cat $OUT/bench.json
timeout 200 python tools/gpu/r3_probe.py --config wholebody --alternate --trace 3 2>&1 | grep -v amdgpu.ids > $OUT/probe_wb.log; tail -62 $OUT/probe_wb.log | head -40
