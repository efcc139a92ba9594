#!/bin/bash
# DSN generator backward on 16-bit shadows: DSN fixtures, DP test, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 1200 python -m pytest tests/test_gpu_dsn.py tests/test_gpu_dp.py -m gpu -x -q -p no:cacheprovider -k "dsn or DSN" > gpurun_out/r03r_pytest.log 2>&1; echo "dsn tests exit $?"; tail -5 gpurun_out/r03r_pytest.log
for v in 1; do
DASR_DSN_BWD16=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03r_bench_$v.json 2> gpurun_out/r03r_bench_$v.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03r_bench_$v.json'))
    print('BWD16=$v: main step %.2f ms'%d['ms_per_step'])
    for s in d.get('secondary',[]):
        if 'DSN' not in s['config']['workload']: continue
        r=s.get('roofline') or {}
        print('  %-60s %.2f ms  frac %s'%(s['config']['workload'][:60], s['ms_per_step'], r.get('frac')))
        for k in (r.get('per_kernel') or [])[:6]: print('       %-56s n=%4d avg=%8.1f us share %.3f %s'%(k['kernel'][:56],k['launches_per_step'],k['avg_launch_us'],k['share_of_kernel_time'],k.get('achieved')))
except Exception as e: print('parse fail', e); print(open('gpurun_out/r03r_bench_$v.err').read()[-800:])
PY
done
echo done
