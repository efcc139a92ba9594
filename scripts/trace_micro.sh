#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -rf gpurun_out/trace_micro
for cin in 16 64 160; do
  timeout 60 python scripts/micro_conv.py --cin $cin --cout 32 --n 16 --reps 40
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_micro/c$cin -o t -- python $R/scripts/micro_conv.py --cin $cin --cout 32 --n 16 --reps 40 > $R/gpurun_out/trace_micro_$cin.log 2>&1)
done
python - <<'PY'
import csv, glob
for cin in (16, 64, 160):
    f = glob.glob('gpurun_out/trace_micro/c%d/*kernel_trace.csv' % cin)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'conv_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    g = [int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) for i in range(len(rows) - 1)]
    g = [x for x in g if x < 100000]
    print('cin %d: n %d  dur avg %.2f us min %.2f  gap avg %.2f us min %.2f' % (cin, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, sum(g) / len(g) / 1e3, min(g) / 1e3))
PY
