#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_steps.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep %.2f ms'%d['ms_per_step'], d['roofline']['kernel_time_over_wall'])"
done
