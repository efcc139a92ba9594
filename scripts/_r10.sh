cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gan.py -m gpu -q -p no:cacheprovider -k "vgg_forward" 2>&1 | grep -aE "passed|failed|Error|^E " | head -20
grep VGG gpurun_out/parity_margins.log
