#!/bin/bash
# round 4, GPU session 4: suite after the split-operand BatchNorm-discriminator weight gradients, the grouped DSN weight gradients and the few-split
# reduce path; bench line with the per-bucket tables
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_c4_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c4_pytest.log
grep -E "passed|failed|FAILED|exit" gpurun_out/r04_c4_pytest.log | tail -12
grep -n "VGG128\|Discriminator_VGG_128\|BatchNorm" gpurun_out/parity_margins.log | cut -c1-400
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r04_c4_bench.json 2> gpurun_out/r04_c4_bench.err
echo "bench exit $?"; tail -2 gpurun_out/r04_c4_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c4_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['kernel_time_over_wall'], d['roofline']['frac'])
for b in d['roofline']['buckets']: print('   ', b)
for s in d.get('secondary',[]):
    print(s.get('ms_per_step'), s.get('config',{}).get('workload','')[:70], s.get('error'), (s.get('roofline') or {}).get('traffic'))
    for b in (s.get('roofline') or {}).get('buckets', []): print('   ', b)
PY
