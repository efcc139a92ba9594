#!/bin/bash
# conv5 (Cout=64, fp32 residual in / fp32 + bf16 out) phase stamps and epilogue sensitivity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for mode in conv5 fwd; do
  for n in 16 8; do
    echo "== mode $mode n $n (product lib)"
    timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n $n --mode $mode --reps 40
    echo "== mode $mode n $n (trace lib)"
    DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n $n --mode $mode --reps 40
  done
done
echo "== conv5 n 16 two streams"
timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n 16 --mode conv5 --reps 40 --streams 2
echo "== conv5 alias 2 (cache-resident)"
timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n 16 --mode conv5 --reps 40 --alias 2
echo "== fwd cout 32 cin 128 n 16 / n 8 trace"
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 120 python scripts/micro_conv.py --cin 128 --cout 32 --n 16 --reps 40
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 120 python scripts/micro_conv.py --cin 128 --cout 32 --n 8 --reps 40
echo done
