cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in 48 8 16 128 48 16; do
DASR_ENQ_CHUNK=$c timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-secondary 2>&1 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chunk $c', d['ms_per_step'], 'kernel_time_over_wall', d['roofline']['kernel_time_over_wall'], 'single', d['roofline'].get('single_stream',{}).get('ms_per_step'))"
done
