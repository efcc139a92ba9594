cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DASR_TUNE="1=14" timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sr.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for mode in fwd dgrad; do
for cin in 64 96 128 160; do
  for t in 12 14; do
    timeout 120 python scripts/micro_conv.py --cin $cin --mode $mode --n 16 --reps 60 --tune 1=$t 2>&1 | tail -1
  done
done
done
for t in 12 14; do
  timeout 120 python scripts/micro_conv.py --cin 128 --mode fwd --n 16 --streams 2 --reps 60 --tune 1=$t 2>&1 | tail -1
done
timeout 300 python bench.py --sweep --sweep-combos "1=12,1=14" --sweep-rounds 2 2>&1 | grep sweep
DASR_STREAMS=1 timeout 300 python bench.py --sweep --sweep-combos "1=12,1=14" --sweep-rounds 2 2>&1 | grep sweep
