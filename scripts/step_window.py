"""What runs inside ONE steady-state production step?  ("186 copyBuffer + bf16 Fill launches per step" -- they are plan-BUILD
work: table uploads and the zero-fill of freshly allocated slabs, divided by the step count.)

Reads the kernel trace of `rocprofv3 --kernel-trace -- python bench.py --steps K --warmup W --no-cpu-baseline --no-secondary`, finds the
adam_kernel dispatches (one per step of the SR trainer), and lists every kernel dispatched between the end of the Adam launch of timed step
i and the end of the Adam launch of timed step i + 1: name, launches, total time.

    python scripts/step_window.py <dir with *_kernel_trace.csv> [index of the step inside the run, default 3] > profiles/r04_step_window.txt
"""
import csv
import glob
import re
import sys
from collections import OrderedDict

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
if len(adam) < which + 2:
    sys.exit('only %d adam launches in the trace' % len(adam))
lo, hi = adam[which], adam[which + 1]
win = rows[lo + 1:hi + 1]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)


acc = OrderedDict()
for s, e, n in win:
    k = short(n)
    a = acc.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
total_k = sum(v[1] for v in acc.values())
print('# one steady-state production step of configs[1] (step %d of the run): wall %.3f ms between the ends of two consecutive Adam launches, '
      '%d kernel dispatches, sum of kernel durations %.3f ms' % (which, (rows[hi][1] - rows[lo][1]) / 1e6, len(win), total_k / 1e6))
print('# whole trace for comparison: %d dispatches, of them __amd_rocclr_copyBuffer %d, at::native fill kernels %d (plan build: table uploads, zero-fill of new slabs)' % (
    len(rows), sum('copyBuffer' in r[2] for r in rows), sum('FillFunctor' in r[2] for r in rows)))
print('%-72s %9s %12s' % ('kernel', 'launches', 'total us'))
foreign = 0
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    own = not ('at::native' in k or 'rocclr' in k)
    foreign += 0 if own else n
    print('%-72s %9d %12.1f%s' % (k[:72], n, t / 1e3, '' if own else '   <- not a dasr kernel'))
print('# dispatches that are not library kernels: %d (the per-step input upload: LR / HR copies into the plan buffers of the two sub-batch replicas)' % foreign)
