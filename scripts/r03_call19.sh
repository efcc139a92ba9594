#!/bin/bash
# store cache policy of the conv epilogues (kernel-boundary write-back of dirty L2 lines) + 4x4 wgrad without spills
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03t_$tag.json 2> gpurun_out/r03t_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r03t_$tag.json')); r=d['roofline']
    print('$tag step %.2f ms %.1f img/s ktime/wall %s'%(d['ms_per_step'], d['value'], r.get('kernel_time_over_wall')))
    for k in r['per_kernel'][:7]: print('   %-50s n=%4d avg=%7.1f us %7.1f TF share %.3f'%(k['kernel'][:50],k['launches_per_step'],k['avg_launch_us'],k['achieved'],k['share_of_kernel_time']))
except Exception as e: print('$tag parse fail', e); print(open('gpurun_out/r03t_$tag.err').read()[-800:])
PY
}
run plain DASR_TUNE=
run sc1 DASR_TUNE=7=1
run sc0sc1 DASR_TUNE=7=2
run nt DASR_TUNE=7=3
run plain_s1 DASR_TUNE= DASR_STREAMS=1
run sc1_s1 DASR_TUNE=7=1 DASR_STREAMS=1
run nt_s1 DASR_TUNE=7=3 DASR_STREAMS=1
echo done
