#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) of the four secondary workloads -> gpurun_out/pmcs_<workload>_<COUNTER>_summary.{txt,json}
# (scripts/collect_profiles.py folds them into profiles/pmc_traffic.json `workloads`)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
run() {  # workload tag, bench args...
  wl=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcs_${wl}_$ctr
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_${wl}_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/pmcs_${wl}_$ctr.log 2>&1)
    echo "pmc $wl $ctr exit $?"
    python scripts/pmc_summary.py gpurun_out/pmcs_${wl}_$ctr $ctr > gpurun_out/pmcs_${wl}_${ctr}_summary.txt 2>&1
    find gpurun_out/pmcs_${wl}_$ctr -type f -size +1M -delete
    head -3 gpurun_out/pmcs_${wl}_${ctr}_summary.txt | cut -c1-170
  done
}
run dasr_vgg --model dasr --fea l1 --batch 32
run dasr_lpips --model dasr --fea LPIPS --batch 32
run dsn_vgg --model dsn --per-type VGG
run dsn_lpips --model dsn --per-type LPIPS
