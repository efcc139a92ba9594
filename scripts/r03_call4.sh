#!/bin/bash
# round 3, GPU session 4: (a) where does the G-gradient error of the VGG128 step case come from (HR tail f16 vs trunk bf16)? (b) wgrad3 ablation series
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for hp in 2 3; do
  rm -f gpurun_out/parity_margins.log
  DASR_HR_PREC=$hp timeout 300 python -m pytest tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider -k "VGG128_gau5" > gpurun_out/r03d_vgg128_hp$hp.log 2>&1
  echo "HR_PREC=$hp exit $?"; grep fp64 gpurun_out/parity_margins.log
done
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03d_$tag.json 2> gpurun_out/r03d_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03d_$tag.json')); r=d['roofline']
    w=[k for k in r['per_kernel'] if k['kernel'].startswith('wgrad3_kernel<true, false, false')]
    print('$tag step %.2f ms; wgrad3:'%d['ms_per_step'], [(k['launches_per_step'], k['avg_launch_us']) for k in w], 'sum us', sum(k['launches_per_step']*k['avg_launch_us'] for k in w))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03d_$tag.err').read()[-800:])
PY
}
export DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_ablate.so DASR_STREAMS=1
for abl in 0 1 2 4 8 3 6 7 9 15; do run abl$abl DASR_WGRAD_ABL=$abl; done
echo done
