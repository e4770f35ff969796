#!/bin/bash
# round 4: fc_chain_kernel with its weight pieces requested twelve at a time and the phase-2 weights touched at the start: parity, cold
# and warm stand-alone times against the previous build, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "fc_chain or chain or mid_models or golden or schedules" > $out/r4fc3_tests.log 2>&1 || { tail -30 $out/r4fc3_tests.log; exit 1; }
tail -2 $out/r4fc3_tests.log
for r in 1 2; do
  for v in fcold base; do
    lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
    echo "== $v (cold)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/_diag/fc_chain_cold.py 2>&1 | grep -v amdgpu
    echo "== $v (warm)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/fc_chain_bench.py 2>&1 | grep -v amdgpu
  done
done > $out/r4_fc3.txt 2>&1
cat $out/r4_fc3.txt
bash tools/ab_bench.sh keisei_amd/libka_fcold.so 8 > $out/r4fc3_ab.txt 2>&1 || { tail -5 $out/r4fc3_ab.txt; exit 1; }
cat $out/r4fc3_ab.txt
