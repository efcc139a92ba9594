#!/bin/bash
# GPU session 2 of round 5: phase offset between the two images of an XCD (stagger sweep + placement trace), DSN one-pass forward parity + timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
timeout 200 python scripts/r05/chain_trace.py --stagger 0,25,45,65,85,110,140 2>&1 | grep -v "INFO\|Warning\|warn" | tee gpurun_out/r05_s2_stagger.log
DASR_HIP_LIB=$R/dasr_amd/libdasr_hip_trace.so timeout 200 python scripts/r05/chain_trace.py --forms 1 2>&1 | grep -v "INFO\|Warning\|warn" | tee gpurun_out/r05_s2_trace.log
rm -f gpurun_out/parity_margins.log
timeout 700 python -m pytest tests/test_gpu_fuzz_shapes.py tests/test_gpu_dsn.py tests/test_gpu_dsn_val.py tests/test_gpu_wgan.py tests/test_gpu_trajectory.py tests/test_gpu_fullsize_steps.py tests/test_gpu_lifetime.py -m gpu -q -p no:cacheprovider -k "dsn or sr_step_on_random or wgan or deterministic" > gpurun_out/r05_s2_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_s2_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exit|Error" gpurun_out/r05_s2_pytest.log | tail -30
grep -i "dsn" gpurun_out/parity_margins.log | tail -30
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05_s2_bench.json 2> gpurun_out/r05_s2_bench.err
echo "bench exit $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s2_bench.json').read().strip().splitlines()[-1])
print('configs[1] ms/step', d['ms_per_step'], 'value', d['value'])
for s in d.get('secondary', []):
    print(s.get('workload', s.get('name')), s.get('ms_per_step'))
PY
for c in 0 1 0 1; do
  DASR_CHAIN=$c timeout 200 python bench.py --model dasr --fea LPIPS --no-cpu-baseline --no-secondary --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[2] LPIPS DASR_CHAIN=$c ms/step', d['ms_per_step'])"
done | tee gpurun_out/r05_s2_gan_chain_ab.log
