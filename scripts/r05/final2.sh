#!/bin/bash
# round 5, last GPU session: the full GPU suite on the committed tree (after the sub-batch chained launches), the bench line as the driver runs it, kernel stats and PMC traffic
# passes of the four secondary workloads again (configs[2] now runs chained launches); the headline's rocprofv3 / PMC artefacts of scripts/r05/final.sh stay valid (same kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/r05g_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05g_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r05g_pytest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -2 gpurun_out/bench.err
bash scripts/prof_secondary.sh > gpurun_out/r05g_secondary.log 2>&1; grep -c kernel_stats gpurun_out/r05g_secondary.log
bash scripts/r05/pmc_secondary.sh > gpurun_out/r05g_pmc_secondary.log 2>&1; grep -c "exit 0" gpurun_out/r05g_pmc_secondary.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))
for s in d.get('secondary',[]): print(s['config']['workload'][:60], s['ms_per_step'], s['roofline']['kernel'], s['roofline']['frac'])
PY
