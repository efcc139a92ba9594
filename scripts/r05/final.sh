#!/bin/bash
# round 5, final GPU session (tag r05f): the full GPU suite on the committed tree, the bench line as the driver runs it, rocprofv3 kernel stats + the two PMC traffic passes of the
# headline (scripts/gpu_round.sh), kernel stats of the four secondary workloads (scripts/prof_secondary.sh), MFMA-busy PMC passes (scripts/pmc_mfma_busy.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r05f_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05f_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r05f_pytest.log | tail -12
RUN_TESTS=0 RUN_BENCH=1 RUN_PROF=1 RUN_PMC=1 PROF_TAG=r05f PROF_STEPS=12 bash scripts/gpu_round.sh > gpurun_out/r05f_round.log 2>&1; tail -4 gpurun_out/r05f_round.log
bash scripts/prof_secondary.sh > gpurun_out/r05f_secondary.log 2>&1; grep -c kernel_stats gpurun_out/r05f_secondary.log
bash scripts/pmc_mfma_busy.sh > gpurun_out/r05f_mfma_busy.log 2>&1; tail -3 gpurun_out/r05f_mfma_busy.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))
for s in d.get('secondary',[]): print(s['config']['workload'][:60], s['ms_per_step'])
PY
