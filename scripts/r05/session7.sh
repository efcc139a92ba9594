#!/bin/bash
# GPU session 7 of round 5: batches of k x 512 tiles as k chained launches over image sub-batches (RRDBNetHIP.chain_split): configs[2] A/B (DASR_CHAIN_SPLIT=1 = exact fit only) + parity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rnd in 1 2; do
  for sp in 1 4; do
    for fea in LPIPS l1; do
      DASR_CHAIN_SPLIT=$sp timeout 200 python bench.py --model dasr --fea $fea --batch 32 --no-cpu-baseline --no-secondary --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[2] fea $fea DASR_CHAIN_SPLIT=$sp ms/step', d['ms_per_step'], d['roofline']['kernel'])"
    done
  done
done | tee gpurun_out/r05_s7_ab.log
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_gan.py tests/test_gpu_fullsize_steps.py tests/test_gpu_dp.py tests/test_gpu_trajectory.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
