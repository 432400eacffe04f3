#!/bin/bash
# Counters of the Winograd kernel (variant $1, default 2) on the four ResNet-50 shapes: separate --pmc passes with --kernel-trace only.
# Output: gpurun_out/wino/pmc_<group>/ + a table on stdout (tools/gpu/winograd_pmc.py).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
REPO="$PWD"; V="${1:-2}"; OUT="$REPO/gpurun_out/wino/pmc_v$V"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); D="$OUT/g$i"
  timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
      python "$REPO/tools/gpu/winograd_probe.py" --timing-only --variants $V --reps 2 > "$D.stdout.log" 2> "$D.stderr.log"
  echo "group $i ($G): rc=$?"
  find "$D" -name '*kernel_trace.csv' -delete
done
cd "$REPO"; python tools/gpu/winograd_pmc.py "$OUT"
