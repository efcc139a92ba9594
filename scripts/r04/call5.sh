#!/bin/bash
# round 4, GPU session 5: f16 storage of the dense blocks (DASR_RDB_PREC=2): parity on the SR / GAN fixtures, the BatchNorm case with its
# automatic selection, same-box step time bf16 vs f16
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 1200 python -m pytest tests/test_gpu_sr.py tests/test_gpu_gan.py tests/test_gpu_lifetime.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider > gpurun_out/r04_c5_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c5_pytest.log
grep -E "passed|failed|FAILED|exit|Error" gpurun_out/r04_c5_pytest.log | tail -15
grep -n "f16\|VGG128\|sr_nf64_nb2_b8_32\|sr_nf64_nb23_b2_32 \|cfg1" gpurun_out/parity_margins.log | cut -c1-420
for prec in 1 2 1 2; do
  DASR_RDB_PREC=$prec timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04_c5_bench_p$prec.json 2> gpurun_out/r04_c5_bench_p$prec.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_c5_bench_p$prec.json')); r=d['roofline']
print('rdb_prec $prec', d['ms_per_step'], d['value'], r['kernel_time_over_wall'], [(k['kernel'][:34], k['avg_launch_us']) for k in r['per_kernel'][:5]])
PY
done
