#!/bin/bash
# round 4, GPU session 7 (final artefact set of the round, tag r04b): full suite (rot_flip, tsamples / tensorboard, BatchNorm-DP fix), steady-state step window, r04 artefact set
# (kernel stats + PMC passes of the headline AND the four secondary workloads)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_c7_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c7_pytest.log
grep -E "passed|failed|FAILED|exit" gpurun_out/r04_c7_pytest.log | tail -12
# ---- steady-state step window
rm -rf gpurun_out/win; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/win -o win -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/win.log 2>&1)
python scripts/step_window.py gpurun_out/win 3 > gpurun_out/r04_step_window.txt 2>&1; head -40 gpurun_out/r04_step_window.txt
rm -rf gpurun_out/win
# ---- headline artefacts: bench line, kernel stats, PMC traffic
RUN_TESTS=0 RUN_BENCH=1 RUN_PROF=1 RUN_PMC=1 PROF_TAG=r04b BENCH_STEPS=8 bash scripts/gpu_round.sh > gpurun_out/r04_round.log 2>&1; tail -5 gpurun_out/r04_round.log
# ---- secondary workloads: kernel stats + PMC traffic
bash scripts/prof_secondary.sh > gpurun_out/r04_prof_secondary.log 2>&1
for wl in "dasr_vgg --model dasr --fea l1 --batch 32" "dasr_lpips --model dasr --fea LPIPS --batch 32" "dsn_vgg --model dsn --per-type VGG" "dsn_lpips --model dsn --per-type LPIPS"; do
  set -- $wl; tag=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcs_${tag}_$ctr
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_${tag}_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/pmcs_${tag}_$ctr.log 2>&1)
    echo "pmc $tag $ctr exit $?"
    python scripts/pmc_summary.py gpurun_out/pmcs_${tag}_$ctr $ctr > gpurun_out/pmcs_${tag}_${ctr}_summary.txt 2>&1
    rm -rf gpurun_out/pmcs_${tag}_$ctr
  done
done
ls gpurun_out | grep -c pmcs_
echo call3 done
