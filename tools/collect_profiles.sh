#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (round 3, re-used since).
# Every profiler run sits under `timeout` (a rocprofv3 that does not exit must not eat the GPU budget), the most
# important outputs come first and the summary is rewritten after every stage, so a run that is cut short still
# leaves what it finished.  Counters are collected in passes of their own with --kernel-trace only.
# Run on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/prof_${ROUND:-r5}; the summaries that are
# judged are copied to profiles/r3/ by hand: rocprofv3_summary.md, *_kernel_stats.csv, pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"
OUT="$REPO/gpurun_out/prof_${ROUND:-r5}"
rm -rf "$OUT"; mkdir -p "$OUT"
STAGES="${STAGES:-1 2 3 4}"        # STAGES=3: only the bench step traces (-> bench_step_summary.md)
has() { [[ " $STAGES " == *" $1 "* ]]; }
export TMPDIR=/tmp
cd /tmp
summarize() {
  (cd "$REPO" && python tools/summarize_profiles.py "$OUT" > "$OUT/rocprofv3_summary.md" 2>&1
   for w in config2 config2_fc config4 config4_fc; do
     f=$(find "$OUT/$w/stats" -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/${w}_kernel_stats.csv"
   done)
}
probe_args() {
  case "$1" in
    config2) echo "--config coco --alternate";;
    config2_fc) echo "--config coco --fc --alternate";;
    config4) echo "--config wholebody --alternate";;
    config4_fc) echo "--config wholebody --fc --alternate";;
  esac
}
# 1. kernel trace + stats of the decode, per workload
has 1 && for w in config2 config4 config2_fc config4_fc; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/stats" -o stats -- \
      python "$REPO/tools/gpu/r3_probe.py" $(probe_args $w) --reps 12 > "$OUT/${w}_stats_stdout.log" 2> "$OUT/${w}_stats_stderr.log"
  find "$OUT/$w/stats" -name '*kernel_trace.csv' -delete
done
summarize
# 2. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
has 2 && for w in config2 config4 config2_fc config4_fc; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$w/pmc_$C" -o pmc -- \
        python "$REPO/tools/gpu/r3_probe.py" $(probe_args $w) --reps 4 > "$OUT/${w}_pmc_${C}_stdout.log" 2> "$OUT/${w}_pmc_${C}_stderr.log"
    find "$OUT/$w/pmc_$C" -name '*kernel_trace.csv' -delete
  done
  summarize
done
# 3. kernel trace of the default bench command's headline leg (N=1, float32 network), and of the bfloat16 leg
if has 3; then
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv -d "$OUT/bench" -o bench -- \
    python "$REPO/bench.py" --steps 8 --warmup 3 --config 2 --no-cpu-baseline --no-parity --no-bf16-leg > "$OUT/bench_stdout.log" 2> "$OUT/bench_stderr.log"
summarize
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/bench_bf16" -o bench -- \
    python "$REPO/bench.py" --steps 8 --warmup 3 --config 2 --no-cpu-baseline --no-parity --backbone-dtype bf16 > "$OUT/bench_bf16_stdout.log" 2> "$OUT/bench_bf16_stderr.log"
summarize
(cd "$REPO" && python tools/summarize_profiles.py "$OUT" --only-steps > "$OUT/bench_step_summary.md" 2>&1)
fi
# 4. cache / LDS / instruction-mix counters of the decode kernels (one small group per pass)
has 4 && for w in config2 config4; do
  for G in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
    D="$OUT/$w/pmc_$(echo $G | cut -d' ' -f1)"
    timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
        python "$REPO/tools/gpu/r3_probe.py" $(probe_args $w) --reps 4 > "$D.stdout.log" 2> "$D.stderr.log"
    find "$D" -name '*kernel_trace.csv' -delete
  done
  summarize
done
# the big traces are not merged back (64 MiB limit): keep the summaries only
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
cd "$REPO"
head -120 "$OUT/rocprofv3_summary.md"
