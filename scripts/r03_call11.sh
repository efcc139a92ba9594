#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03k_$tag.json 2> gpurun_out/r03k_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03k_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if 'conv_glds_kernel<1' in k['kernel']]
    print('$tag step %.2f ms;'%d['ms_per_step'], [(k['kernel'][17:40], k['launches_per_step'], k['avg_launch_us']) for k in w])
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03k_$tag.err').read()[-500:])
PY
}
export DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_ablate.so
run abl0 DASR_TUNE=1=100
run abl1_nodma DASR_TUNE=1=101
run abl4_nowait_nobarrier DASR_TUNE=1=104
run abl16_barrier_nowait DASR_TUNE=1=116
run abl17_nodma_barrier DASR_TUNE=1=117
run abl8_nomfma DASR_TUNE=1=108
echo done
