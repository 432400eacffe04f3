#!/bin/bash
# phase timing of the growers (diagnostic library: python -c "from openpifpaf_amd import build; build.build_diagnostic('OPA_ASSOC_PHASE_TIMING', 'ph')")
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for cfg in "${@:-wholebody}"; do
  echo "=== $cfg"
  OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_ph.so timeout 300 python tools/gpu/r3_probe.py --config $cfg --reps 2 2>&1 | grep -v amdgpu.ids > gpurun_out/phase_raw.log
  grep -v PHASE gpurun_out/phase_raw.log | grep -v "^ \|^img\|status" | head -6
  grep -a "PHASE img" gpurun_out/phase_raw.log | tail -64 | python -c "
import sys, re
names={0:'pop+entry reads',1:'query+issue loads',2:'wait loads',3:'window test/compact',4:'score (exp)',5:'top-2 reductions',6:'blend finish',7:'connection rest',8:'re-push',9:'assign',10:'publish box',11:'frontier_add_from',12:'task setup+seed publish',13:'(grow end)',14:'pose boxes+score',15:'idle/wait task',16:'(stamp cost)',17:'to scan start',20:'walk: gather jobs',21:'walk: start values',22:'batch: query+chunk boxes',23:'batch: items',24:'batch: loads+window+compact',25:'batch: scores (exp)',26:'batch: top-2',27:'batch: targets+finish',28:'walk: results+candidates',29:'walk: publish',30:'walk: poll'}
rows={}
for ln in sys.stdin:
    m=re.search(r'PHASE img (\\d+) k (\\d+) cycles (\\d+) n (\\d+)', ln)
    if not m: continue
    img,k,cyc,n=map(int,m.groups()); rows.setdefault(img,[]).append((k,cyc,n))
for img,r in rows.items():
    tot=sum(c for k,c,n in r if k not in (15,))
    print('image %d: total grower cycles (excl. idle) %d' % (img,tot))
    for k,c,n in r:
        if n: print('  %2d %-24s %10d cycles %5.1f%%  n %6d  %7.0f cycles each' % (k,names.get(k,'?'),c,100.0*c/max(1,tot),n,c/max(1,n)))
"
done
