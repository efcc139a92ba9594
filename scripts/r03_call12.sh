#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
DASR_TUNE=${T:-1=15} timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_steps.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03l_pytest.log 2>&1; echo "pytest($T) exit $?"; tail -2 gpurun_out/r03l_pytest.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03l_$tag.json 2> gpurun_out/r03l_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03l_$tag.json')); r=d['roofline']
    print('$tag step %.2f ms %.1f img/s ktime/wall %s'%(d['ms_per_step'], d['value'], r.get('kernel_time_over_wall')))
    for k in r['per_kernel'][:7]: print('   %-50s n=%4d avg=%7.1f us %7.1f TF share %.3f'%(k['kernel'][:50],k['launches_per_step'],k['avg_launch_us'],k['achieved'],k['share_of_kernel_time']))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03l_$tag.err').read()[-800:])
PY
}
run base DASR_TUNE=
run ring DASR_TUNE=${T:-1=15}
run ring_s1 DASR_TUNE=${T:-1=15} DASR_STREAMS=1
run base_s1 DASR_TUNE= DASR_STREAMS=1
echo done
