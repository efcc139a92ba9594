for v in ${ABLS:-0 1 2 4 8 16 3 11 27}; do
  if [ $v = 0 ]; then L=dasr_amd/libdasr_hip.so; else L=dasr_amd/libdasr_hip_isabl$v.so; fi
  echo "IS_ABL=$v"; DASR_ALLOW_NONFINITE=1 DASR_HIP_LIB=$L timeout 200 python scripts/r06/is_trace.py 2>&1 | grep -E "chain" | grep -v check
done
