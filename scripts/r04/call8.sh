#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
free -g | head -2; nproc
rm -f gpurun_out/parity_margins.log
( time timeout 900 python -m pytest tests/test_gpu_fullsize_steps.py tests/test_gpu_dsn.py -m gpu -q -p no:cacheprovider -k "cfg4_exact" --durations=5 ) 2>&1 | tail -22
cat gpurun_out/parity_margins.log | cut -c1-400
