#!/bin/bash
# round 4: fc_chain_kernel phase-1 unroll depth (weight pieces in flight): 4 (shipped) / 8 / 12, one box, alternating
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for r in 1 2; do
  for v in base fcu8 fcu12; do
    lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
    echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/fc_chain_bench.py 2>&1 | grep -v amdgpu
  done
done > $out/r4_fc_unroll.txt 2>&1
cat $out/r4_fc_unroll.txt
