#!/bin/bash
# compile-time ablations of rdb_is_kernel (csrc/rdb_is.h IS_ABL bits, WRONG results, timing only) as separate libraries dasr_amd/libdasr_hip_isabl<bits>.so:
#   bash scripts/r06/is_ablate.sh 1 2 4 ...   then   DASR_HIP_LIB=dasr_amd/libdasr_hip_isabl1.so python scripts/r06/is_trace.py
set -u
cd "$(dirname "$0")/../.."
python -m dasr_amd.build > /dev/null 2>&1
B=dasr_amd/build
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${IS_EXTRA:-} -DIS_ABL=$v -c dasr_amd/csrc/conv.hip -o $B/conv_isabl$v.o > /dev/null 2>&1 \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dasr_amd/libdasr_hip_isabl$v.so $B/conv_isabl$v.o $B/wgrad.o $B/misc.o $B/gan.o $B/lpips.o $B/rccl.o -ldl -pthread \
    && echo "built isabl$v" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
