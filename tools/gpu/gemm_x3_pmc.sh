#!/bin/bash
# Counters of the split-operand GEMM (csrc/gemm_f32x3.hip) on the layer-3 reduce shape: separate --pmc passes with --kernel-trace only.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
REPO="$PWD"; OUT="$REPO/gpurun_out/x3pmc"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); D="$OUT/g$i"
  SHAPES="layer3 reduce" timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$D" -o pmc -- \
      python "$REPO/tools/gpu/gemm_x3_probe.py" > "$D.stdout.log" 2> "$D.stderr.log"
  echo "group $i ($G): rc=$?"
  find "$D" -name '*kernel_trace.csv' -delete
done
cd "$REPO"; python - "$OUT" <<'P'
import collections, csv, glob, os, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], 'g*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if 'gemm_f32' not in n:
            continue
        key = n.split('(')[0][-60:]
        rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(rows):
    print(key)
    for name in sorted(rows[key]):
        v = rows[key][name]
        print('    %-36s %16.1f   (%d launches)' % (name, sum(v) / len(v), len(v)))
P
