#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; export TMPDIR=/tmp
rm -rf gpurun_out/prof_dsn
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dsn -o dsn -- python $R/bench.py --model dsn --per-type VGG --steps 10 --warmup 2 > $R/gpurun_out/prof_dsn.log 2>&1)
find gpurun_out/prof_dsn -name "*kernel_trace*" -delete
head -25 gpurun_out/prof_dsn/dsn_kernel_stats.csv | cut -c1-170
