#!/bin/bash
# round 3, GPU session 9: the artefacts of the round: bench line, rocprofv3 kernel stats (production schedule and single stream), PMC traffic passes, MFMA-busy passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
TAG=${TAG:-r03i}
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -c 600 gpurun_out/bench.log | head -c 300; echo
rm -rf gpurun_out/prof gpurun_out/prof_ss
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o $TAG -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof.log 2>&1); echo "prof exit $?"
(cd /tmp && DASR_STREAMS=1 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ss -o ${TAG}_ss -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_ss.log 2>&1); echo "prof_ss exit $?"
find gpurun_out/prof gpurun_out/prof_ss -name "*kernel_trace*" -size +20M -delete
grep '^{"metric"' gpurun_out/prof.log | head -c 400 > gpurun_out/${TAG}_bench_under_rocprof.json
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$ctr.log 2>&1)
  echo "pmc $ctr exit $?"
  python scripts/pmc_summary.py gpurun_out/pmc_$ctr $ctr > gpurun_out/pmc_${ctr}_summary.txt 2>&1
  find gpurun_out/pmc_$ctr -type f -size +1M -delete
done
bash scripts/pmc_mfma_busy.sh > gpurun_out/pmc_busy.log 2>&1; echo "mfma busy exit $?"
python scripts/pmc_mfma_busy.py gpurun_out > gpurun_out/${TAG}_pmc_mfma_busy.txt 2>&1; head -16 gpurun_out/${TAG}_pmc_mfma_busy.txt | cut -c1-150
find gpurun_out/prof gpurun_out/prof_ss -name "*stats*" | head
echo done
