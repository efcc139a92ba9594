cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for z in 0 1; do
for t in 100 107 115; do
for cin in 64 160; do
    timeout 120 python scripts/micro_conv.py --cin $cin --mode fwd --n 16 --reps 100 --tune 1=$t --zero $z 2>&1 | tail -1
done
done
done
