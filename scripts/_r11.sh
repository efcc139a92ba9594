cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -E "sclk|mclk|Power|power" | head -8
for z in 0 1; do
( timeout 60 python scripts/micro_conv.py --cin 160 --mode fwd --n 16 --reps 40000 --tune 1=107 --zero $z > /tmp/mc_$z.log 2>&1 & )
sleep 9
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Average Graphics Package Power|Socket Power" | tr '\n' ' '; echo; sleep 0.7; done
sleep 14
tail -1 /tmp/mc_$z.log
done
( timeout 60 python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/b.log 2>&1 & )
sleep 12
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Average Graphics Package Power|Socket Power" | tr '\n' ' '; echo; sleep 0.7; done
sleep 12
