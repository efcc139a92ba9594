#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for kv in 0 1; do
export HIP_FORCE_DEV_KERNARG=$kv
echo "HIP_FORCE_DEV_KERNARG=$kv"
timeout 120 python scripts/micro_conv.py --cin 64 --cout 32 --n 16 --reps 60 --mode fwd 2>&1 | tail -1
timeout 120 python scripts/micro_conv.py --cin 160 --cout 32 --n 16 --reps 60 --mode fwd 2>&1 | tail -1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', j['ms_per_step'])"
done
