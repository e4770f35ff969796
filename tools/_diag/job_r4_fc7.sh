#!/bin/bash
# fc_chain_kernel with the LDS pieces of a batch read up front: parity, per-wave stamps, cold / warm stand-alone, step A/B vs the previous build
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "fc_chain or chain or mid_models or golden or schedules" > $out/r4fc7_tests.log 2>&1 || { tail -30 $out/r4fc7_tests.log; exit 1; }
tail -1 $out/r4fc7_tests.log
KEISEI_AMD_LIB=$PWD/keisei_amd/libka_fctl.so timeout -k 10 200 python tools/_diag/fc_chain_tl.py 2>&1 | grep -v amdgpu | grep -A1 -E "gpool fwd|gpool bwd" | cut -c1-250
for r in 1 2; do
  for v in fcold base; do
    lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
    echo "== $v (cold)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/_diag/fc_chain_cold.py 2>&1 | grep -v amdgpu
    echo "== $v (warm)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/fc_chain_bench.py 2>&1 | grep -v amdgpu
  done
done > $out/r4_fc7.txt 2>&1
cat $out/r4_fc7.txt
bash tools/ab_bench.sh keisei_amd/libka_fcold.so 8 > $out/r4fc7_ab.txt 2>&1 || { tail -5 $out/r4fc7_ab.txt; exit 1; }
cat $out/r4fc7_ab.txt
