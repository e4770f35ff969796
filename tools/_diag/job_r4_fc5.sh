#!/bin/bash
# round 4: fc_chain_kernel with four accumulators per tile (+ batched weight pieces, phase-2 weights touched at the start)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/r4fc5_tests.log 2>&1 || { tail -30 $out/r4fc5_tests.log; exit 1; }
tail -2 $out/r4fc5_tests.log
KEISEI_AMD_LIB=keisei_amd/libka_fctl.so timeout -k 10 200 python tools/_diag/fc_chain_tl.py 2>&1 | grep -v amdgpu > $out/r4_fc_tl3.txt; cat $out/r4_fc_tl3.txt | cut -c1-260
for r in 1 2; do
  for v in fcold base; do
    lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
    echo "== $v (cold)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/_diag/fc_chain_cold.py 2>&1 | grep -v amdgpu
    echo "== $v (warm)"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/fc_chain_bench.py 2>&1 | grep -v amdgpu
  done
done > $out/r4_fc5.txt 2>&1
cat $out/r4_fc5.txt
bash tools/ab_bench.sh keisei_amd/libka_fcold.so 8 > $out/r4fc5_ab.txt 2>&1 || { tail -5 $out/r4fc5_ab.txt; exit 1; }
cat $out/r4fc5_ab.txt
