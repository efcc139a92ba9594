#!/bin/bash
# GPU session 1 of round 5: full parity suite in the new order, chain phase trace (both forms), same-box A/B of the chain forms, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r05_s1_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_s1_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r05_s1_pytest.log | tail -30
cp gpurun_out/parity_margins.log gpurun_out/r05_s1_parity_margins.log 2>/dev/null
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 200 python scripts/r05/chain_trace.py > gpurun_out/r05_s1_chain_trace.log 2>&1
echo "trace exit $?" >> gpurun_out/r05_s1_chain_trace.log
tail -40 gpurun_out/r05_s1_chain_trace.log
for rnd in 1 2; do
  DASR_TUNE=7=1 timeout 120 python scripts/r04/step_time.py --label "chain form 1 (round 4)" 2>&1 | tail -1
  DASR_TUNE=7=2 timeout 120 python scripts/r04/step_time.py --label "chain form 2 (round 5)" 2>&1 | tail -1
done | tee gpurun_out/r05_s1_ab.log
timeout 600 python bench.py > gpurun_out/r05_s1_bench.json 2> gpurun_out/r05_s1_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r05_s1_bench.err; tail -c 2500 gpurun_out/r05_s1_bench.json
