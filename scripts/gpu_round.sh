#!/bin/bash
# One GPU-box session: parity tests, variant sweep, bench line, rocprof kernel stats (+ optional PMC passes).
# Outputs under gpurun_out/.  Every command is bounded by `timeout` and reads stdin from /dev/null.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
exec < /dev/null
if [ "${RUN_TESTS:-1}" = "1" ]; then
  rm -f gpurun_out/parity_margins.log
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest.log
  grep -E "passed|failed|error|Error|assert|FAILED|exit" gpurun_out/pytest.log | tail -40
fi
if [ "${RUN_SWEEP:-0}" = "1" ]; then
  timeout 400 python bench.py --sweep --sweep-combos "${SWEEP_COMBOS:-6=0,6=1}" --sweep-rounds ${SWEEP_ROUNDS:-3} > gpurun_out/sweep.log 2>&1
  echo "sweep exit $?" >> gpurun_out/sweep.log
  grep -E "sweep|exit|Error|error" gpurun_out/sweep.log | tail -40
fi
if [ "${RUN_BENCH:-1}" = "1" ]; then
  timeout ${BENCH_TIMEOUT:-600} python bench.py --steps ${BENCH_STEPS:-8} --warmup 2 ${BENCH_ARGS:-} > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench exit $?" >> gpurun_out/bench.err
  tail -5 gpurun_out/bench.err; tail -c 6000 gpurun_out/bench.log
fi
if [ "${RUN_PROF:-0}" = "1" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o ${PROF_TAG:-r02} -- python $R/bench.py --steps ${PROF_STEPS:-4} --warmup 1 --no-cpu-baseline --no-secondary ${PROF_ARGS:-} > $R/gpurun_out/prof.log 2>&1)
  echo "prof exit $?" >> gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
  find gpurun_out/prof -type f | head
  tail -c 3000 gpurun_out/prof.log
fi
if [ "${RUN_PMC:-0}" = "1" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$ctr
    (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary ${PMC_ARGS:-} > $R/gpurun_out/pmc_$ctr.log 2>&1)
    echo "pmc $ctr exit $?"
    python scripts/pmc_summary.py gpurun_out/pmc_$ctr $ctr > gpurun_out/pmc_${ctr}_summary.txt 2>&1
    find gpurun_out/pmc_$ctr -type f -size +1M -delete
  done
fi
if [ -n "${EXTRA_CMD:-}" ]; then
  bash -c "$EXTRA_CMD"
fi
echo "gpu_round done"
