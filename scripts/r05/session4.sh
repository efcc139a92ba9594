#!/bin/bash
# GPU session 4 of round 5: conv5 of a chained dense block publishes its 16-bit shadow before the fp32 stream is stored (CHV=128); reduce with one workgroup per output channel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rnd in 1 2 3; do
  timeout 100 python scripts/r04/step_time.py --label "product (reduce: 32 workgroups per part)" 2>&1 | tail -1
  DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v128.so timeout 100 python scripts/r04/step_time.py --label "CHV=128 (conv5: shadow, flag, then fp32)" 2>&1 | tail -1
done | tee gpurun_out/r05_s4_ab.log
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v128.so timeout 300 python -m pytest tests/test_gpu_sr.py tests/test_gpu_fullsize_steps.py tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider -k "chain or cfg1" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_dsn.py tests/test_gpu_fullsize_steps.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
