#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sr or dp or fullsize" 2>&1 | tail -2
timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', j['ms_per_step'], 'single', j['roofline']['single_stream']['ms_per_step'], 'peak', j['roofline']['peak_at_observed_clock'], [s.get('ms_per_step') for s in j['secondary']])
for r in j['roofline']['per_kernel'][:8]: print('2s', r['kernel'], r['avg_launch_us'], r['frac'])
for r in j['roofline']['single_stream']['per_kernel'][:8]: print('1s', r['kernel'], r['avg_launch_us'], r['frac'])"
