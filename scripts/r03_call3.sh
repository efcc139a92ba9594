#!/bin/bash
# round 3, GPU session 3: the whole -m gpu suite after the round's host-side changes, then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r03c_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" gpurun_out/r03c_pytest.log | tail -5; grep -E "^(FAILED|ERROR)" gpurun_out/r03c_pytest.log | head -20
grep -A18 "slowest" gpurun_out/r03c_pytest.log | head -24
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c_bench.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], d.get('mfma_util_step'), 'frac', r['frac'], r['kernel'], 'single', r.get('frac_single_stream'))
for s in d.get('secondary',[]): print(s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('error'))
print(d.get('cpu_baseline'))
PY
echo done
