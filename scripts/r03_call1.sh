#!/bin/bash
# round 3, GPU session 1: baseline of the production schedule, stream-count A/B, SQ counter pass (LDS / issue stalls) on single-stream launches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for s in 2 4 1; do
  DASR_STREAMS=$s timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03a_bench_s$s.json 2> gpurun_out/r03a_bench_s$s.err
  echo "streams $s exit $?"; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03a_bench_s$s.json')); print('streams $s', d['ms_per_step'], d['value'], d['roofline'].get('kernel_time_over_wall'), d['roofline'].get('mfma_only_tflops_by_operand_data'))
except Exception as e: print('parse fail', e)
PY
done
for ctrs in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL"; do
  tag=$(echo $ctrs | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && DASR_STREAMS=1 timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$tag.log 2>&1)
  echo "pmc $tag exit $?"
  for c in $ctrs; do python scripts/pmc_summary.py gpurun_out/pmc_$tag $c > gpurun_out/r03a_sq_${c}.txt 2>&1; cp gpurun_out/pmc_${tag}_summary.json gpurun_out/r03a_sq_${c}.json; done
  find gpurun_out/pmc_$tag -type f -size +1M -delete
done
python - <<'PY'
import json,glob
names=['SQ_BUSY_CYCLES','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_LDS','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VMEM','SQ_INST_CYCLES_VMEM','SQ_INSTS_LDS','SQ_VALU_MFMA_BUSY_CYCLES','SQ_WAVES','SQ_LDS_DATA_FIFO_FULL']
d={}
for n in names:
    try: d[n]=json.load(open('gpurun_out/r03a_sq_%s.json'%n))
    except Exception as e: print('missing',n,e)
ks=[k for k in d.get('SQ_BUSY_CYCLES',{}) if k.startswith(('conv_glds','wgrad3','wgrad_reduce','conv_kernel'))]
print('%-46s'%'kernel'+' '.join('%14s'%n[3:17] for n in names if n in d))
for k in ks:
    print('%-46s'%k[:46]+' '.join('%14.0f'%(d[n].get(k,{}).get('avg_kb',-1)) for n in names if n in d))
PY
echo done
