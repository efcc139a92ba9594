cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for mode in fwd dgrad conv5; do
for cin in 64 160; do
    co=32; if [ $mode = conv5 ]; then co=64; cin=192; fi
    timeout 120 python scripts/micro_conv.py --cin $cin --cout $co --mode $mode --n 16 --reps 100 2>&1 | tail -1
done
done
timeout 120 python scripts/micro_conv.py --cin 64 --mode fwd --n 16 --reps 100 --tune 1=115 2>&1 | tail -1
timeout 300 python bench.py --steps 8 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
DASR_STREAMS=1 timeout 300 python bench.py --steps 8 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
