cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sr.py -m gpu -q -p no:cacheprovider -k "ps_nf64" 2>&1 | grep -aE "passed|failed|FAILED|Error" | tail
grep -aE "sr_ps" gpurun_out/parity_margins.log
