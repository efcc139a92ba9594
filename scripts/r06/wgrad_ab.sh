#!/bin/bash
# round 6: the register-window weight-gradient kernel -- correctness tests on the product library, then the same-process A/B on the ablation library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_dsn.py -m gpu -x -q -p no:cacheprovider -k "wgrad or grad or step or iteration or trainer" > gpurun_out/wgrad_ab_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/wgrad_ab_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/wgrad_ab_pytest.log | tail -6
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_ablate.so timeout 600 python scripts/r06/wgrad_ab.py 2>&1 | grep -v Warning | tail -8 | tee gpurun_out/wgrad_ab.log
