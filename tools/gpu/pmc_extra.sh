#!/bin/bash
# only the cache / LDS / instruction-mix passes of tools/collect_profiles.sh (reuses an existing gpurun_out/prof_r1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT="$PWD/gpurun_out/prof_r1"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for G in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  D="$OUT/pmc_$(echo $G | cut -d' ' -f1)"
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
      python "$OLDPWD/bench.py" --decode-only --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline > "$D.stdout.log" 2> "$D.stderr.log"
  echo "$G: rc=$?"
done
cd "$OLDPWD"
find "$OUT" -name "*counter_collection.csv" | head
