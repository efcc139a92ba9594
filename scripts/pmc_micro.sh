#!/bin/bash
# SQ counters of the isolated dense conv (batch 16, Cin 160 -> 32) -> gpurun_out/pmc_micro/ ; two passes of 8 SQ slots each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
ARGS="${MICRO_ARGS:---cin 160 --cout 32 --n 16}"
rm -rf gpurun_out/pmc_micro
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_micro/p$i -o pmc -- python $R/scripts/micro_conv.py $ARGS --reps 6 > $R/gpurun_out/pmc_micro_$i.log 2>&1)
  echo "pmc pass $i exit $?"
done
find gpurun_out/pmc_micro -name "*kernel_trace*" -delete
