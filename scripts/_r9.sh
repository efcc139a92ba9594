cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider 2>&1 | grep -aE "passed|failed|FAILED" | tail -3
timeout 300 python bench.py --sweep --sweep-combos "2=13,2=12" --sweep-rounds 2 2>&1 | grep sweep
DASR_STREAMS=4 timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
DASR_STREAMS=4 DASR_TUNE="2=12" timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
