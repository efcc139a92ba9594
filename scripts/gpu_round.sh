#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest.log
tail -40 gpurun_out/pytest.log
if [ "${RUN_BENCH:-1}" = "1" ]; then
  timeout 200 python bench.py --steps 2 --warmup 1 --nb 2 --batch 2 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
  echo "bench_small exit $?" >> gpurun_out/bench_small.log; tail -4 gpurun_out/bench_small.log
  timeout ${BENCH_TIMEOUT:-420} python bench.py --steps ${BENCH_STEPS:-4} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1)
  echo "prof exit $?" >> gpurun_out/prof.log
  find gpurun_out/prof -name "*stats*" | head
  # keep only the small summaries (traces can be large)
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
