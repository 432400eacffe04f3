#!/bin/bash
# round 5 (second session): why did force complete on the field itself (RAW lists, commit 55f1dc7) lose to materialised lists
# (f0ad6e9)?  Both trees are built under _fcdiag/{raw,mat}; per-kernel HIP-event times and PMC passes of the same workload.
cd "${GRAFT_REPO_ROOT:-.}"
REPO="$PWD"; OUT="$REPO/gpurun_out/fcdiag"; mkdir -p "$OUT"
export TMPDIR=/tmp
for v in mat raw; do
  cd "$REPO/_fcdiag/$v"
  for cfg in "coco --fc --alternate" ; do
    echo "=== $v: $cfg"; timeout 200 python tools/gpu/r3_probe.py --config $cfg --reps 20 2>&1 | grep -E "cif_active|wall"
  done
  cd /tmp
  for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    D="$OUT/$v/pmc_$(echo $G | cut -d' ' -f1)"
    mkdir -p "$D"; timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
        python "$REPO/_fcdiag/$v/tools/gpu/r3_probe.py" --config coco --fc --alternate --reps 4 > "$D.stdout.log" 2> "$D.stderr.log"
    find "$D" -name '*kernel_trace.csv' -delete
  done
done
cd "$REPO"
python tools/gpu/fcdiag_summary.py "$OUT" | tee "$OUT/summary.txt"
