#!/bin/bash
# round 4, GPU session 11: full GPU suite of the final tree (parity margins -> profiles/r04_parity_margins.log) + the bench line as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r04_c11_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c11_pytest.log
grep -E "passed|failed|FAILED|exit" gpurun_out/r04_c11_pytest.log | tail -12
grep -A18 "slowest" gpurun_out/r04_c11_pytest.log | head -20
timeout 600 python bench.py > gpurun_out/r04_c11_bench.json 2> gpurun_out/r04_c11_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r04_c11_bench.err; head -c 1500 gpurun_out/r04_c11_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_c11_bench.json') if l.startswith('{')][-1])
print('\nheadline', d['ms_per_step'], d['value'], 'zero', d['roofline'].get('zero_operand_step'))
for s in d.get('secondary',[]): print(s['config']['workload'][:60], s['ms_per_step'])
PY
