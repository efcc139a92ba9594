#!/bin/bash
# compile-time variants of the form-1 chain loop (conv.hip -DCHV=bits) as separate libraries: dasr_amd/libdasr_hip_v<bits>.so
set -u
cd "$(dirname "$0")/../.."
python -m dasr_amd.build > /dev/null 2>&1
B=dasr_amd/build
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCHV=$v -c dasr_amd/csrc/conv.hip -o $B/conv_v$v.o > /dev/null 2>&1 \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dasr_amd/libdasr_hip_v$v.so $B/conv_v$v.o $B/wgrad.o $B/misc.o $B/gan.o $B/lpips.o $B/rccl.o -ldl -pthread \
    && echo "built v$v" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
