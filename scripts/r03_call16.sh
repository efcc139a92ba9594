#!/bin/bash
# split 16-bit tensors: kernel tests, VGG test, GAN / DSN step fixtures, secondary bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "split or conv5 or forward_matches" > gpurun_out/r03q_pytest_k.log 2>&1; echo "kernels exit $?"; tail -5 gpurun_out/r03q_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider -k "vgg_forward" > gpurun_out/r03q_pytest_v.log 2>&1; echo "vgg exit $?"; tail -8 gpurun_out/r03q_pytest_v.log
if [ "${FULL:-0}" = "1" ]; then
timeout 1500 python -m pytest tests/test_gpu_gan.py tests/test_gpu_dsn.py tests/test_gpu_fullsize_gan.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03q_pytest_g.log 2>&1; echo "gan+dsn exit $?"; tail -5 gpurun_out/r03q_pytest_g.log
for pv in 5; do
DASR_VGG_PREC=$pv timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03q_bench_p$pv.json 2> gpurun_out/r03q_bench_p$pv.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03q_bench_p$pv.json'))
    print('VGG prec $pv: main step %.2f ms'%d['ms_per_step'])
    for s in d.get('secondary',[]):
        r=s.get('roofline') or {}
        print('  %-60s %.2f ms  frac %s'%(s['config']['workload'][:60], s['ms_per_step'], r.get('frac')))
        for k in (r.get('per_kernel') or [])[:5]: print('       %-56s n=%4d avg=%8.1f us share %.3f %s'%(k['kernel'][:56],k['launches_per_step'],k['avg_launch_us'],k['share_of_kernel_time'],k.get('achieved')))
except Exception as e: print('parse fail', e); print(open('gpurun_out/r03q_bench_p$pv.err').read()[-800:])
PY
done
fi
echo done
