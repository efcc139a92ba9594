"""Per-kernel average of a rocprofv3 --pmc counter from its counter_collection CSV (written on the GPU box; the raw CSV is too big to pull)."""
import csv
import glob
import sys
from collections import defaultdict

d, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get('Counter_Name') != ctr:
            continue
        a = acc[row['Kernel_Name']]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
print('kernel,dispatches,avg_%s' % ctr)
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.1f' % (k, n, v / n))
