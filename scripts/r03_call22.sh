#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rep in 1 2; do
for c in 48 12 24 128; do
  DASR_ENQ_CHUNK=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep chunk $c %.2f ms'%d['ms_per_step'], d['roofline']['kernel_time_over_wall'])"
done
done
