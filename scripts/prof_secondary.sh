#!/bin/bash
# kernel stats of the secondary workloads (configs[2] GAN step with VGG / LPIPS, configs[4] DSN iteration with VGG / LPIPS):
# one rocprofv3 --kernel-trace --stats run each; the stats CSVs land in gpurun_out/prof_sec/ (copy into profiles/ by hand)
R=${GRAFT_REPO_ROOT:-.}
cd $R; export TMPDIR=/tmp
rm -rf gpurun_out/prof_sec; mkdir -p gpurun_out/prof_sec
run() {  # tag, bench args...
    tag=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sec -o $tag -- \
        python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/prof_sec/$tag.log 2>&1)
    tail -1 gpurun_out/prof_sec/$tag.log | cut -c1-200
}
run dasr_vgg --model dasr --fea l1 --batch 32
run dasr_lpips --model dasr --fea LPIPS --batch 32
run dsn_vgg --model dsn --per-type VGG
run dsn_lpips --model dsn --per-type LPIPS
find gpurun_out/prof_sec -name "*kernel_trace*" -delete
find gpurun_out/prof_sec -name "*_kernel_stats.csv" | while read f; do echo "== $f"; head -14 $f | cut -c1-150; done
