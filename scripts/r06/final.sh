#!/bin/bash
# round 6, final GPU session (tag r06g): the full GPU suite on the committed tree (-x, as the driver runs it), smoke, the bench line as the driver runs it, rocprofv3 kernel stats +
# the two PMC traffic passes of the headline (scripts/gpu_round.sh), kernel stats of the four secondary workloads (scripts/prof_secondary.sh), MFMA-busy PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/r06g_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r06g_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r06g_pytest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
RUN_TESTS=0 RUN_BENCH=1 RUN_PROF=1 RUN_PMC=1 PROF_TAG=r06g PROF_STEPS=12 BENCH_TIMEOUT=900 bash scripts/gpu_round.sh > gpurun_out/r06g_round.log 2>&1; tail -4 gpurun_out/r06g_round.log
bash scripts/prof_secondary.sh > gpurun_out/r06g_secondary.log 2>&1; grep -c kernel_stats gpurun_out/r06g_secondary.log
bash scripts/r05/pmc_secondary.sh > gpurun_out/r06g_pmc_secondary.log 2>&1; grep -c "exit 0" gpurun_out/r06g_pmc_secondary.log
bash scripts/pmc_mfma_busy.sh > gpurun_out/r06g_mfma_busy.log 2>&1; tail -3 gpurun_out/r06g_mfma_busy.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['avg_launch_us'], r['frac'], r.get('frac_of_peak_on_data'), 'cpu', d.get('cpu_baseline',{}).get('value'))
for s in d.get('secondary',[]): print(s.get('config',{}).get('workload','?')[:70], s.get('ms_per_step'), s.get('roofline',{}).get('kernel'), s.get('roofline',{}).get('frac'), s.get('error'))
PY
