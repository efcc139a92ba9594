cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest.log
grep -aE "passed|failed|FAILED|exit|Error" gpurun_out/pytest.log | tail -12
timeout 400 python bench.py --steps 8 > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['ms_per_step'], d['value'], d.get('cpu_baseline',{}).get('value'))
for r in d.get('secondary',[]): print(r.get('metric'), r.get('value'), r.get('ms_per_step'), r.get('error'))
PY
