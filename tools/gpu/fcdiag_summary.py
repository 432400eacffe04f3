"""Per-kernel averages of the PMC passes tools/gpu/r5b_fcdiag.sh collected (rocprofv3 counter_collection CSVs)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernels = ('cifcaf_fc_kernel', 'cafscored_kernel', 'cifcaf_assoc_kernel')
for v in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, v)):
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, v, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                k = next((k for k in kernels if k in name), None)
                if k is None:
                    continue
                a = acc[(k, row['Counter_Name'])]
                a[0] += float(row['Counter_Value']); a[1] += 1
    print('== %s' % v)
    for (k, c), (s, n) in sorted(acc.items()):
        print('%-22s %-22s launches %3d  mean %.4g' % (k, c, n, s / max(n, 1)))
