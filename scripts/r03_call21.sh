#!/bin/bash
# same-box A/B of two library builds: dasr_amd/libdasr_hip_prev.so (previous commit) vs dasr_amd/libdasr_hip.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rep in 1 2 3; do
for lib in prev cur; do
  if [ $lib = prev ]; then export DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_prev.so; else unset DASR_HIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep $lib %.2f ms'%d['ms_per_step'], d['roofline']['kernel_time_over_wall'])"
done
done
