#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 600 python -m pytest tests/test_gpu_sr.py -m gpu -x -q -p no:cacheprovider -k "nb2_b8 or nb23_b2_32" > gpurun_out/r03h_pytest.log 2>&1; echo "pytest(w4) exit $?"; tail -2 gpurun_out/r03h_pytest.log
DASR_WGRAD4=0 DASR_WGRAD_GLDS=1 timeout 600 python -m pytest tests/test_gpu_sr.py -m gpu -x -q -p no:cacheprovider -k "nb2_b8 or nb23_b2_32" > gpurun_out/r03h_pytest2.log 2>&1; echo "pytest(glds) exit $?"; tail -2 gpurun_out/r03h_pytest2.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03h_$tag.json 2> gpurun_out/r03h_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03h_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if 'wgrad' in k['kernel']]
    print('$tag step %.2f ms; wgrad:'%d['ms_per_step'], [(k['kernel'][:34], k['launches_per_step'], k['avg_launch_us'], k['achieved']) for k in w])
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03h_$tag.err').read()[-600:])
PY
}
run w3 DASR_WGRAD4=0
run w3glds DASR_WGRAD4=0 DASR_WGRAD_GLDS=1
run w4 DASR_WGRAD4=1
run w3glds_g4 DASR_WGRAD4=0 DASR_WGRAD_GLDS=1 DASR_WG_GROUP=4
run w3_g4 DASR_WGRAD4=0 DASR_WG_GROUP=4
echo done
