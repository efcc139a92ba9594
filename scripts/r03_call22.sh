#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rep in 1 2 3; do
for a in 0 1; do
  DASR_ABL_MASK=$a timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep mask-ablation $a %.2f ms'%d['ms_per_step'])"
done
done
