#!/bin/bash
# round 4, GPU session 12 (final, tag r04d): full GPU suite of the tree with the chained trunk launches (parity margins -> profiles/r04_parity_margins.log), the bench line
# as the driver runs it, rocprofv3 kernel stats and the two PMC traffic passes of the headline (the chain kernels are new names in all of them)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r04_c12_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c12_pytest.log
grep -E "passed|failed|FAILED|exit" gpurun_out/r04_c12_pytest.log | tail -12
timeout 600 python bench.py > gpurun_out/r04_c12_bench.json 2> gpurun_out/r04_c12_bench.err
echo "bench exit $?"; tail -2 gpurun_out/r04_c12_bench.err
RUN_TESTS=0 RUN_BENCH=0 RUN_PROF=1 RUN_PMC=1 PROF_TAG=r04d bash scripts/gpu_round.sh > gpurun_out/r04d_round.log 2>&1; tail -4 gpurun_out/r04d_round.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_c12_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'zero', r.get('zero_operand_step'))
for s in d.get('secondary',[]): print(s['config']['workload'][:60], s['ms_per_step'])
PY
