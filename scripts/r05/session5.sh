#!/bin/bash
# GPU session 5 of round 5: balanced grouped weight-gradient launches (dasr_wgrad_map) A/B + parity; --wgan x --ragan / data parallel; full suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
for rnd in 1 2 3; do
  DASR_WGRAD_BALANCE=0 timeout 100 python scripts/r04/step_time.py --label "uniform splits (round 3: 12 dense blocks x 20 workgroups)" 2>&1 | tail -1
  DASR_WGRAD_BALANCE=1 timeout 100 python scripts/r04/step_time.py --label "balanced splits (14 dense blocks x 18 workgroups)" 2>&1 | tail -1
done | tee gpurun_out/r05_s5_ab.log
rm -f gpurun_out/parity_margins.log
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > gpurun_out/r05_s5_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_s5_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit" gpurun_out/r05_s5_pytest.log | tail -30
cp gpurun_out/parity_margins.log gpurun_out/r05_s5_parity_margins.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r05_s5_bench.json 2> gpurun_out/r05_s5_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s5_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', d['ms_per_step'], d['value'], r['kernel'], r['frac'])
for s in d.get('secondary',[]): print(s['config']['workload'][:70], s['ms_per_step'])
PY
