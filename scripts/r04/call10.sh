#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 900 python -m pytest tests/test_gpu_wgan.py tests/test_gpu_dsn.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40
grep -n "wgan" gpurun_out/parity_margins.log | cut -c1-300
