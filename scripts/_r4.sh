cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.log
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_kernels.py tests/test_gpu_infer.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
grep -E "^sr_|^cfg1" gpurun_out/parity_margins.log | grep -E "activations|gradients"
for m in f32 f16; do
DASR_HR_STORE=$m timeout 300 python bench.py --steps 8 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done
timeout 300 python bench.py --steps 8 --no-cpu-baseline --no-secondary > gpurun_out/bench_f16.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_f16.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['ms_per_step'], d['value'])
for r in d['roofline']['per_kernel']: print(r['kernel'], r['launches_per_step'], r['avg_launch_us'], r['achieved'], r['share_of_kernel_time'])
PY
