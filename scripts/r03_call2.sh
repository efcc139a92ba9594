#!/bin/bash
# round 3, GPU session 2: deferred / grouped dense-block weight gradients: parity tests, then A/B of the schedules on the bench workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_dp.py tests/test_gpu_fullsize_steps.py tests/test_gpu_gan.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03b_pytest.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/r03b_pytest.log
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03b_bench_$tag.json 2> gpurun_out/r03b_bench_$tag.err
  echo "$tag exit $?"; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03b_bench_$tag.json')); r=d['roofline']
    print('$tag', d['ms_per_step'], d['value'], 'ktime/wall', r.get('kernel_time_over_wall'))
    for k in r['per_kernel'][:8]: print('   %-46s n=%4d avg=%7.1f us %7.1f TF share %.3f'%(k['kernel'][:46],k['launches_per_step'],k['avg_launch_us'],k['achieved'],k['share_of_kernel_time']))
except Exception as e: print('parse fail', e); print(open('gpurun_out/r03b_bench_$tag.err').read()[-1500:])
PY
}
run defer0 DASR_WG_DEFER=0
run defer1 DASR_WG_DEFER=1
run defer1_g1 DASR_WG_DEFER=1 DASR_WG_GROUP=1
run defer1_g4 DASR_WG_DEFER=1 DASR_WG_GROUP=4
run defer1_glds DASR_WG_DEFER=1 DASR_WGRAD_GLDS=1
run defer1_s1 DASR_WG_DEFER=1 DASR_STREAMS=1
echo done
