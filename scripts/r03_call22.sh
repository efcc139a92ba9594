#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rep in 1 2; do
for g in 4 2 8 12 23; do
  DASR_WG_GROUP=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep group $g %.2f ms'%d['ms_per_step'])"
done
done
