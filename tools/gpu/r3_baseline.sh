#!/bin/bash
# Round 3, first GPU call: the suites at HEAD, then statistics + rocprofv3 kernel tables of the two shapes round 2
# left un-profiled (wholebody batch 16; COCO with the reference benchmark's force-complete setting).
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"; OUT="$REPO/gpurun_out/r3_base"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee "$OUT/tests.log"
for cfg in "--config coco" "--config coco --fc" "--config wholebody" "--config wholebody --fc" "--config coco --alternate"; do
  echo "=== $cfg"; timeout 300 python tools/gpu/r3_probe.py $cfg --check 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/probe.log"
done
cd /tmp
for name in "wholebody:--config wholebody" "coco_fc:--config coco --fc" "wholebody_fc:--config wholebody --fc"; do
  tag="${name%%:*}"; a="${name#*:}"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -o $tag -- \
      python "$REPO/tools/gpu/r3_probe.py" $a --reps 10 > "$OUT/${tag}_stdout.log" 2> "$OUT/${tag}_stderr.log"
  f=$(find "$OUT/$tag" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/${tag}_kernel_stats.csv" && head -12 "$f"
  find "$OUT/$tag" -name '*kernel_trace.csv' -delete
done
