#!/bin/bash
# round 3, GPU session 7: wgrad4 ablation series (libdasr_hip_ablate.so), single stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03g_$tag.json 2> gpurun_out/r03g_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03g_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if 'wgrad4<false' in k['kernel'] or k['kernel'].startswith('wgrad3_kernel<true, false, false')]
    print('$tag step %.2f ms; wgrad:'%d['ms_per_step'], [(k['kernel'][:28], k['launches_per_step'], k['avg_launch_us']) for k in w])
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03g_$tag.err').read()[-600:])
PY
}
export DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_ablate.so DASR_STREAMS=1 DASR_WGRAD4=1
for abl in 0 1 2 4 8 3 6 9 11 15; do run abl$abl DASR_WGRAD_ABL=$abl; done
echo done
