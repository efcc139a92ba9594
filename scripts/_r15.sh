cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --model dasr --batch 32 --steps 4 2>/dev/null > gpurun_out/bench_dasr.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_dasr.log') if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['kernel_time_over_wall'], d['roofline']['non_mfma_kernel_time_share'])
for r in d['roofline']['per_kernel']:
    print('%-64s n %4d avg %8.1f us  %7.1f TF  share %.3f'%(r['kernel'][:64], r['launches_per_step'], r['avg_launch_us'], r['achieved'], r['share_of_kernel_time']))
PY
