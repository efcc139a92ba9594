#!/bin/bash
# DSN generator on split f16 tensors / 16-bit backward: DSN fixtures, DP test, full-size batch split, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 1200 python -m pytest tests/test_gpu_dsn.py tests/test_gpu_dp.py tests/test_gpu_fullsize_steps.py tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "dsn or DSN or split or conv5" > gpurun_out/r03r_pytest.log 2>&1; echo "dsn tests exit $?"; tail -5 gpurun_out/r03r_pytest.log
for v in 1 0; do
DASR_DSN_FWD16=$v timeout 600 python bench.py --model dsn --per-type LPIPS --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03r_bench_$v.json 2> gpurun_out/r03r_bench_$v.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03r_bench_$v.json')); r=d.get('roofline') or {}
    print('FWD16=$v: DSN step %.2f ms  %.1f crops/s mfma_util_step %s'%(d['ms_per_step'], d['value'], r.get('mfma_util_step')))
    for k in (r.get('per_kernel') or [])[:6]: print('       %-56s n=%4d avg=%8.1f us share %.3f %s'%(k['kernel'][:56],k['launches_per_step'],k['avg_launch_us'],k['share_of_kernel_time'],k.get('achieved')))
except Exception as e: print('parse fail', e); print(open('gpurun_out/r03r_bench_$v.err').read()[-800:])
PY
done
grep "DSN" gpurun_out/parity_margins.log | tail -12
echo done
