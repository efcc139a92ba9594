#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for cin in 96 160; do
for st in 1 2; do
  for m in "fwd 0" "dgrad 0" "dgrad 3"; do
    set -- $m
    timeout 120 python scripts/micro_conv.py --cin $cin --cout 32 --n 16 --mode $1 --alias $2 --reps 60 --streams $st 2>&1 | grep -v amdgpu.ids
  done
done
done
