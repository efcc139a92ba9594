#!/bin/bash
# wide fp32 epilogue accesses (v_permlane16_swap): parity + conv5 micro + step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_steps.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03o_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03o_pytest.log
for n in 16 8; do
  timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n $n --mode conv5 --reps 40
done
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n 16 --mode conv5 --reps 40
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03o_$tag.json 2> gpurun_out/r03o_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03o_$tag.json')); r=d['roofline']
    print('$tag step %.2f ms %.1f img/s ktime/wall %s'%(d['ms_per_step'], d['value'], r.get('kernel_time_over_wall')))
    for k in r['per_kernel'][:7]: print('   %-50s n=%4d avg=%7.1f us %7.1f TF share %.3f'%(k['kernel'][:50],k['launches_per_step'],k['avg_launch_us'],k['achieved'],k['share_of_kernel_time']))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03o_$tag.err').read()[-800:])
PY
}
run base
run s1 DASR_STREAMS=1
run t12 DASR_TUNE=2=12
echo done
