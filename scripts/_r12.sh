cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sr.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider 2>&1 | grep -aE " passed| failed|FAILED|^E " | tail -5
for b in 1 3 1 3 6; do
DASR_WG_BATCH=$b timeout 300 python bench.py --steps 8 --no-cpu-baseline --no-secondary 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/wg_batch $b /"
done
