cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m dasr_amd.build --trace > /dev/null 2>&1
for t in 12 115; do
for cin in 64 160; do
    DASR_HIP_LIB=$GRAFT_REPO_ROOT/dasr_amd/libdasr_hip_trace.so timeout 120 python scripts/micro_conv.py --cin $cin --mode fwd --n 16 --reps 50 --tune 1=$t 2>&1 | tail -16
done
done
