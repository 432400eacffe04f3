#!/bin/bash
# SQ / instruction-cache counters of the association kernel (one rocprofv3 --pmc pass per group, --kernel-trace only).
#   bash tools/gpu/r3_sq_counters.sh [probe args...]      -> gpurun_out/sq/<group>.txt (per-counter mean over the launches)
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/sq"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---config coco}"
i=0
for G in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1)); D="$OUT/g$i"; rm -rf "$D"
  timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
      python "$REPO/tools/gpu/r3_probe.py" $ARGS --reps 4 > "$OUT/g${i}_stdout.log" 2> "$OUT/g${i}_stderr.log"
  python - "$D" <<'P' | tee "$OUT/g$i.txt"
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cifcaf_assoc_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print('%-32s %16.0f  (mean of %d launches)' % (k, s / n, n))
P
  rm -rf "$D"
done
