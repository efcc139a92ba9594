#!/bin/bash
# round 3, GPU session 5: wgrad3 ablation series with compile-time variants (libdasr_hip_ablate.so), single stream, deferred grouped launches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03e_$tag.json 2> gpurun_out/r03e_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03e_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if k['kernel'].startswith('wgrad3_kernel<true, false, false')]
    print('$tag step %.2f ms; wgrad3:'%d['ms_per_step'], [(k['launches_per_step'], k['avg_launch_us']) for k in w], 'sum us', sum(k['launches_per_step']*k['avg_launch_us'] for k in w))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03e_$tag.err').read()[-600:])
PY
}
export DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_ablate.so DASR_STREAMS=1
for abl in 0 1 2 4 8 6 7 9 14 15 0; do run abl$abl DASR_WGRAD_ABL=$abl; done
echo done
