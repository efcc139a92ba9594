#!/bin/bash
# round 5: full GPU suite (-x, as the driver runs it), smoke, bench line on the last commit that touches dasr_amd/ or tests/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05h_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05h_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r05h_pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))
for s in d.get('secondary',[]): print(s['config']['workload'][:60], s['ms_per_step'], s['roofline']['kernel'], s['roofline']['frac'])
PY
