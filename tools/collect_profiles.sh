#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (round 2).
# Run on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/prof_r2; the summaries that are
# judged are copied to profiles/r2/ by hand: summary.md, decode_kernel_stats.csv, pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"
OUT="$REPO/gpurun_out/prof_r2"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace + stats of the default bench command's headline leg (N=1, float32 network)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-leg > "$OUT/bench_stdout.log" 2> "$OUT/bench_stderr.log"
# 1b. the same with the network in bfloat16
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_bf16" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --backbone-dtype bf16 > "$OUT/bench_bf16_stdout.log" 2> "$OUT/bench_bf16_stderr.log"
# 2. decode only (no backbone): the hot path's kernels in isolation
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/decode" -o decode -- \
    python "$REPO/bench.py" --decode-only --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/decode_stdout.log" 2> "$OUT/decode_stderr.log"
# 3. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE cannot share a pass); counters only with
#    --kernel-trace (no sys/hip/hsa trace domains)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
      python "$REPO/bench.py" --decode-only --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline > "$OUT/pmc_${C}_stdout.log" 2> "$OUT/pmc_${C}_stderr.log"
done
# 4. cache / LDS / instruction-mix counters of the decode kernels (one small group per pass)
for G in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  D="$OUT/pmc_$(echo $G | cut -d' ' -f1)"
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
      python "$REPO/bench.py" --decode-only --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline > "$D.stdout.log" 2> "$D.stderr.log"
done
cd "$REPO"
python tools/summarize_profiles.py "$OUT" > "$OUT/summary.md" 2>&1
cp "$(find "$OUT/decode" -name '*kernel_stats.csv' | head -1)" "$OUT/decode_kernel_stats.csv" 2>/dev/null
head -90 "$OUT/summary.md"
