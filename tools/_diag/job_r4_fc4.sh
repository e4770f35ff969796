#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "fc_chain or chain or mid_models or golden or schedules" > $out/r4fc4_tests.log 2>&1 || { tail -30 $out/r4fc4_tests.log; exit 1; }
tail -2 $out/r4fc4_tests.log
KEISEI_AMD_LIB=keisei_amd/libka_fctl.so timeout -k 10 200 python tools/_diag/fc_chain_tl.py 2>&1 | grep -v amdgpu > $out/r4_fc_tl2.txt; cat $out/r4_fc_tl2.txt | cut -c1-260
for r in 1 2; do
  for v in fcold base; do
    lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
    echo "== $v (cold)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/_diag/fc_chain_cold.py 2>&1 | grep -v amdgpu
  done
done > $out/r4_fc4.txt 2>&1
cat $out/r4_fc4.txt
bash tools/ab_bench.sh keisei_amd/libka_fcold.so 8 > $out/r4fc4_ab.txt 2>&1 || { tail -5 $out/r4fc4_ab.txt; exit 1; }
cat $out/r4fc4_ab.txt
