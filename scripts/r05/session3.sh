#!/bin/bash
# GPU session 3 of round 5: compile-time variants of the form-1 chain loop (scripts/r05/build_variants.sh), whole configs[1] step, same box, two rounds; DSN / fuzz parity re-run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rnd in 1 2; do
  timeout 100 python scripts/r04/step_time.py --label "CHV=0 (product)" 2>&1 | tail -1
  for v in 1 2 3 4 7 8 16 32 64; do
    DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_v$v.so timeout 100 python scripts/r04/step_time.py --label "CHV=$v" 2>&1 | tail -1
  done
done | tee gpurun_out/r05_s3_variants.log
rm -f gpurun_out/parity_margins.log
timeout 700 python -m pytest tests/test_gpu_fuzz_shapes.py tests/test_gpu_dsn.py tests/test_gpu_dsn_val.py tests/test_gpu_wgan.py tests/test_gpu_trajectory.py tests/test_gpu_fullsize_steps.py tests/test_gpu_sr.py -m gpu -q -p no:cacheprovider -k "dsn or sr_step_on_random or wgan or chain" > gpurun_out/r05_s3_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_s3_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit|Error" gpurun_out/r05_s3_pytest.log | tail -30
grep -i "one_f16_pass\|configs\[4\]" gpurun_out/parity_margins.log | tail
