#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for al in "0 0" "0 1" "0 2" "0 3" "2 3" "0 4" "1 0"; do
  set -- $al
  timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n 16 --mode conv5 --reps 40 --alias $1 --alias5 $2 2>&1 | grep -v amdgpu.ids
done
echo "-- zero data"
timeout 120 python scripts/micro_conv.py --cin 192 --cout 64 --n 16 --mode conv5 --reps 40 --zero 1 2>&1 | grep -v amdgpu.ids
echo "-- cin 64 (4 chunks) conv5: the epilogue share grows"
timeout 120 python scripts/micro_conv.py --cin 64 --cout 64 --n 16 --mode conv5 --reps 40 2>&1 | grep -v amdgpu.ids
timeout 120 python scripts/micro_conv.py --cin 64 --cout 64 --n 16 --mode conv5 --reps 40 --alias 2 --alias5 3 2>&1 | grep -v amdgpu.ids
echo done
