#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rep in 1 2; do
for st in 2 4; do
  DASR_STREAMS=$st timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>gpurun_out/r03u_$st.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep streams $st %.2f ms'%d['ms_per_step'], d['roofline'].get('kernel_time_over_wall'))" || tail -3 gpurun_out/r03u_$st.err
done
done
