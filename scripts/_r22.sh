#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for pv in 3 4; do
echo "DASR_D_PREC=$pv"; rm -f gpurun_out/parity_margins.log
DASR_D_PREC=$pv timeout 600 python -m pytest tests/test_gpu_gan.py tests/test_gpu_dsn.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
grep "worst gradient" gpurun_out/parity_margins.log | cut -c1-110
done
