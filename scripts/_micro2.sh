cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for t in 12 100 101 102 103 104 107 108 112 115; do
for cin in 64 160; do
    timeout 120 python scripts/micro_conv.py --cin $cin --mode fwd --n 16 --reps 100 --tune 1=$t 2>&1 | tail -1
done
done
