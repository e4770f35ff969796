#!/bin/bash
# round 4, job N: what the per-board barrier of the weight-gradient kernel costs (ablation build: a barrier every fourth board, wrong results)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for rep in 1 2; do
for v in base wgnobar; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib MFMA_ONE=fwd2,wgrad,wgradf MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -2
done
done > $out/r4n_wgrad_barrier_ablation.txt 2>&1
cat $out/r4n_wgrad_barrier_ablation.txt
