#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (round 3, re-used since).
# Every profiler run sits under `timeout` (a rocprofv3 that does not exit must not eat the GPU budget), the most
# important outputs come first and the summary is rewritten after every stage, so a run that is cut short still
# leaves what it finished.  Counters are collected in passes of their own with --kernel-trace only.
# Run on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/prof_${ROUND:-r6}; the summaries that are
# judged are copied to profiles/r3/ by hand: rocprofv3_summary.md, *_kernel_stats.csv, pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"
OUT="$REPO/gpurun_out/prof_${ROUND:-r6}"
rm -rf "$OUT"; mkdir -p "$OUT"
STAGES="${STAGES:-1 2 3 4 5}"        # STAGES=3: only the bench step traces (-> bench_step_summary.md)
has() { [[ " $STAGES " == *" $1 "* ]]; }
export TMPDIR=/tmp
cd /tmp
summarize() {
  (cd "$REPO" && python tools/summarize_profiles.py "$OUT" > "$OUT/rocprofv3_summary.md" 2>&1
   for w in config2 config2_fc config4 config4_fc config2_b256; do
     f=$(find "$OUT/$w/stats" -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/${w}_kernel_stats.csv"
   done)
}
probe_args() {
  case "$1" in
    # (--bench-batches: the two field batches bench.py alternates, so that these averages and bench.py's HIP-event times describe
    # the same launches -- round 5 profiled seeds 0 / 1000 and benched seeds 0 / 100000: 2.51 vs 2.77 ms for the wholebody association)
    config2) echo "--config coco --alternate --bench-batches";;
    config2_fc) echo "--config coco --fc --alternate --bench-batches";;
    config4) echo "--config wholebody --alternate --bench-batches";;
    config4_fc) echo "--config wholebody --fc --alternate --bench-batches";;
    config2_b256) echo "--config coco --batch 256 --alternate --bench-batches";;
  esac
}
# 1. kernel trace + stats of the decode, per workload
has 1 && for w in config2 config2_b256 config4 config2_fc config4_fc; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/stats" -o stats -- \
      python "$REPO/tools/gpu/r3_probe.py" $(probe_args $w) --reps 12 > "$OUT/${w}_stats_stdout.log" 2> "$OUT/${w}_stats_stderr.log"
  find "$OUT/$w/stats" -name '*kernel_trace.csv' -delete
done
summarize
# 2. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
has 2 && for w in config2 config2_b256 config4 config2_fc config4_fc; do
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
has 4 && for w in config2 config4 config2_b256; do
  for G in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
    D="$OUT/$w/pmc_$(echo $G | cut -d' ' -f1)"
    timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
        python "$REPO/tools/gpu/r3_probe.py" $(probe_args $w) --reps 4 > "$D.stdout.log" 2> "$D.stderr.log"
    find "$D" -name '*kernel_trace.csv' -delete
  done
  summarize
done
# 5. calibration of FETCH_SIZE on known byte counts in this path's access shapes (tools/gpu/micro/readbw.hip): 4-byte and 16-byte
#    loads per lane, plane-strided reads of 7 planes in 8, on a buffer beyond the Infinity Cache (256 images) and inside it (32)
if has 5 && [ -x "$REPO/tools/gpu/micro/readbw" ]; then
for n in 256 32; do
  D="$OUT/calib_$n"
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D" -o pmc -- "$REPO/tools/gpu/micro/readbw" $n > "$D.stdout.log" 2> "$D.stderr.log"
  f=$(find "$D" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" $n > "$OUT/fetch_size_calibration_$n.md" <<'PY'
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get('Counter_Name') == 'FETCH_SIZE']
images = int(sys.argv[2]); total = images * 19 * 8 * 6561 * 4
acc = defaultdict(list)
for r in rows:
    acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
print('# FETCH_SIZE calibration, buffer of %d images = %.1f MB (readbw.hip; every kernel reads the whole buffer, the plane kernels 7/8 of it)\n' % (images, total / 1e6))
print('| kernel | launches | FETCH_SIZE KiB (mean) | bytes read / (FETCH_SIZE x 1024) |')
print('|---|---|---|---|')
for k, v in acc.items():
    m = sum(v) / len(v)
    read = total * (7 / 8 if 'planes' in k else 1.0)
    print('| `%s` | %d | %.0f | %.2f |' % (k[:60], len(v), m, read / (m * 1024) if m else float('nan')))
PY
done
fi
# the big traces are not merged back (64 MiB limit): keep the summaries only
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
cd "$REPO"
head -120 "$OUT/rocprofv3_summary.md"
