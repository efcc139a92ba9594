#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
DASR_RDB_PREC=2 timeout 300 python scripts/r04/debug_f16.py sr_nf64_nb2_b2_32 2>&1 | grep -v Warn | tail -48
timeout 600 python -m pytest tests/test_gpu_sr.py tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider -k "f16 or VGG128 or srcVGG" 2>&1 | tail -15; grep -n "f16\|VGG128" gpurun_out/parity_margins.log | cut -c1-330
