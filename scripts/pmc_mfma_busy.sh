#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R; export TMPDIR=/tmp
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $ctrs | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && DASR_STREAMS=1 timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$tag.log 2>&1)
  echo "pmc $tag exit $?"
  for c in $ctrs; do python scripts/pmc_summary.py gpurun_out/pmc_$tag $c > gpurun_out/pmc_sq_${c}.txt 2>&1; cp gpurun_out/pmc_${tag}_summary.json gpurun_out/pmc_sq_${c}.json; head -9 gpurun_out/pmc_sq_${c}.txt | cut -c1-200; done
  find gpurun_out/pmc_$tag -type f -size +1M -delete
done
