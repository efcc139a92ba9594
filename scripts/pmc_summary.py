"""Per-kernel average of a rocprofv3 --pmc counter from its counter_collection CSV (run on the GPU box; the raw CSV is too big to pull).
python scripts/pmc_summary.py <dir> <COUNTER>  ->  text lines  +  <dir>/../pmc_<COUNTER>.json ({kernel name as bench.py prints it: avg KB})"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

d, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get('Counter_Name') != ctr:
            continue
        acc[row['Kernel_Name']].append(float(row['Counter_Value']))


def short(n):   # "void (anonymous namespace)::conv_glds_kernel<1, 67, 4, 0, false>(dasr_conv_params)" -> "conv_glds_kernel<1, 67, 4, 0, false>"
    m = re.search(r'(\w+<[^>]*>)\(', n) or re.search(r'::(\w+)\(', n) or re.search(r'(\w+)\(', n)
    return m.group(1) if m else n


out = {}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-10s %-100s launches %5d  avg %.1f KB  min %.1f  max %.1f' % (ctr, k[:100], len(v), sum(v) / len(v), min(v), max(v)))
    out[short(k)] = {'launches': len(v), 'avg_kb': sum(v) / len(v)}
json.dump(out, open(d.rstrip('/') + '_summary.json', 'w'), indent=1)
