cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest.log
grep -aE "passed|failed|FAILED|exit|Error" gpurun_out/pytest.log | tail -12
grep -aE "sr_ps|FSD-Batch" gpurun_out/parity_margins.log
