#!/bin/bash
# round 3, GPU session 6: wgrad4_kernel: parity (SR steps incl. nb23 @128^2, HR tail f16, full-size property tests), then A/B against wgrad3 on the bench step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_steps.py tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03f_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r03f_pytest.log; grep "gradients" gpurun_out/parity_margins.log | cut -c1-160
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03f_$tag.json 2> gpurun_out/r03f_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03f_$tag.json')); r=d['roofline']
    print('$tag step %.2f ms %.1f img/s ktime/wall %s'%(d['ms_per_step'], d['value'], r.get('kernel_time_over_wall')))
    for k in r['per_kernel'][:8]: print('   %-46s n=%4d avg=%7.1f us %7.1f TF share %.3f'%(k['kernel'][:46],k['launches_per_step'],k['avg_launch_us'],k['achieved'],k['share_of_kernel_time']))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03f_$tag.err').read()[-800:])
PY
}
run w4 DASR_WGRAD4=1
run w3 DASR_WGRAD4=0
run w4_s1 DASR_WGRAD4=1 DASR_STREAMS=1
run w4_g4 DASR_WGRAD4=1 DASR_WG_GROUP=4
echo done
