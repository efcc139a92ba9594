#!/bin/bash
# GPU session 6 of round 5: -DCHV=128 = plain (write-back) stores of the fp32 stream in a chained conv5 (read again only by the same tile) vs write-through
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rnd in 1 2 3; do
  timeout 100 python scripts/r04/step_time.py --label "product (every chained store write-through)" 2>&1 | tail -1
  DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v128.so timeout 100 python scripts/r04/step_time.py --label "CHV=128 (conv5 fp32 stream: plain stores)" 2>&1 | tail -1
done | tee gpurun_out/r05_s6_ab.log
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v128.so timeout 400 python -m pytest tests/test_gpu_sr.py tests/test_gpu_fullsize_steps.py tests/test_gpu_gan.py tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -k "chain or cfg1" 2>&1 | tail -3
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v128.so timeout 300 python scripts/r04/chain_soak.py --steps 150 --more 300 2>&1 | tail -6
