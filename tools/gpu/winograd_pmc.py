"""Table of the counters tools/gpu/winograd_pmc.sh collected: one row per (Winograd launch shape = grid size), mean per launch."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'g*', '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if 'winograd' not in r.get('Kernel_Name', ''):
                continue
            key = int(r.get('Grid_Size', 0)) // max(1, int(r.get('Workgroup_Size', 1)))
            rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(rows, reverse=True):
    print('workgroups %d' % key)
    for name in sorted(rows[key]):
        v = rows[key][name]
        print('    %-44s %16.1f   (%d launches)' % (name, sum(v) / len(v), len(v)))
