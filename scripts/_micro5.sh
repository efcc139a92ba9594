cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for kv in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do
for t in 115 12; do
    env $kv timeout 120 python scripts/micro_conv.py --cin 64 --mode fwd --n 16 --reps 100 --tune 1=$t 2>&1 | tail -1 | sed "s/^/$kv /"
done
done
env HIP_FORCE_DEV_KERNARG=0 timeout 200 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
env HIP_FORCE_DEV_KERNARG=1 timeout 200 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
