#!/bin/bash
# same-box A/B (round 4): is the fabric (MALL / HBM) read traffic what bounds the dense-block convs under the two-stream schedule?
# libdasr_hip_ablate.so, DASR_TUNE 1=164: activations read from a cache-resident 256 KB window (wrong results, same instruction stream).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export DASR_HIP_LIB=$PWD/dasr_amd/libdasr_hip_ablate.so DASR_ALLOW_NONFINITE=1
out=gpurun_out/r04_alias_ab.txt
: > $out
for rnd in 1 2; do
  for cfg in "1=12" "1=164" "1=101" "1=108"; do
    for st in 2 1; do
      DASR_STREAMS=$st DASR_TUNE=$cfg timeout 200 python scripts/r04/step_time.py --label "round $rnd streams $st tune $cfg" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
    done
  done
done
cat $out
