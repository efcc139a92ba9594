#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_steps.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03j_pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03j_pytest.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03j_$tag.json 2> gpurun_out/r03j_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03j_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if 'wgrad' in k['kernel']]
    print('$tag step %.2f ms; wgrad:'%d['ms_per_step'], [(k['kernel'][:30], k['launches_per_step'], k['avg_launch_us'], k['achieved']) for k in w])
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03j_$tag.err').read()[-600:])
PY
}
run ld1 DASR_WGRAD_LD=1
run ld0 DASR_WGRAD_LD=0
run ld1_g16 DASR_WGRAD_LD=1 DASR_WG_GROUP=16
run ld1_g1 DASR_WGRAD_LD=1 DASR_WG_GROUP=1
run ld1_s1 DASR_WGRAD_LD=1 DASR_STREAMS=1
echo done
