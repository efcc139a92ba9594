#!/bin/bash
# round 4, GPU session 2: full -m gpu suite after the library pruning / knob collapse (+ world-8, BatchNorm-DP, HR_PREC=3 tests), default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_c2_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c2_pytest.log
grep -E "passed|failed|FAILED|Error|exit" gpurun_out/r04_c2_pytest.log | tail -30
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r04_c2_bench.json 2> gpurun_out/r04_c2_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r04_c2_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c2_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['kernel_time_over_wall'], d['roofline']['frac'])
for s in d.get('secondary',[]): print(s.get('ms_per_step'), s.get('config',{}).get('workload','')[:60], s.get('error'))
print(d.get('cpu_baseline'))
PY
