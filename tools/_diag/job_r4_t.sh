#!/bin/bash
# round 4, job T: wgrad_flat staging through buffer descriptors (branch-free, 15 vector instructions per board instead of ~85)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/wgrad_lean_check.py > $out/r4t_check.txt 2>&1 || { tail -5 $out/r4t_check.txt; exit 1; }
grep -v amdgpu.ids $out/r4t_check.txt | tail -8
KEISEI_AMD_LIB=keisei_amd/libka_wgtl.so timeout -k 10 200 python tools/_diag/wgrad_tl.py > $out/r4t_wgrad_tl.txt 2>&1 || { tail -5 $out/r4t_wgrad_tl.txt; exit 1; }
cat $out/r4t_wgrad_tl.txt
for r in 1 2; do
  echo "== previous build"; KEISEI_AMD_LIB=keisei_amd/libka_old.so MFMA_ONE=wgrad,wgradf MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep wgrad
  echo "== this build"; MFMA_ONE=wgrad,wgradf MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep wgrad
done > $out/r4t_standalone.txt 2>&1
cat $out/r4t_standalone.txt
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "wgrad" > $out/r4t_tests.log 2>&1 || { tail -20 $out/r4t_tests.log; exit 1; }
tail -2 $out/r4t_tests.log
bash tools/ab_bench.sh keisei_amd/libka_old.so 8 > $out/r4t_ab.txt 2>&1 || { tail -5 $out/r4t_ab.txt; exit 1; }
cat $out/r4t_ab.txt
