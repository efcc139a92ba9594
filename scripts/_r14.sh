cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>/dev/null > gpurun_out/bench_ss.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_ss.log') if l.startswith('{')][-1])
print(d['ms_per_step'])
ss=d['roofline']['single_stream']
print('single', ss['ms_per_step'])
tot=0
for r in ss['per_kernel']:
    print('%-60s n %4d avg %8.1f us  %7.1f TF  share %.3f'%(r['kernel'][:60], r['launches_per_step'], r['avg_launch_us'], r['achieved'], r['share_of_kernel_time']))
PY
for b in 1 3; do
DASR_STREAMS=1 DASR_WG_BATCH=$b timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams1 wg_batch $b', d['ms_per_step'])"
done
