cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.log
timeout 900 python -m pytest tests/test_gpu_gan.py tests/test_gpu_dsn.py tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest.log
grep -aE "passed|failed|FAILED|exit|Error|assert" gpurun_out/pytest.log | tail -15
for v in 3 2; do
DASR_VGG_PREC=$v timeout 300 python bench.py --model dasr --batch 32 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dasr vgg prec $v', d['ms_per_step'], d['value'])"
DASR_VGG_PREC=$v timeout 300 python bench.py --model dsn --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dsn vgg prec $v', d['ms_per_step'], d['value'])"
done
