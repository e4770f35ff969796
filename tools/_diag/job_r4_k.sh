#!/bin/bash
# round 4, job K: masked epilogue variants, all built with -fno-slp-vectorize: noslp = the pipelined pair epilogue, varB = + the weight ring
# kept across it, varC = the per-board lean epilogue; base = the product build (SLP on)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for rep in 1 2; do
for v in base noslp varB varC; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib MFMA_ONE=fwd2,dgrad,dgradm MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -3
done
done > $out/r4k_variants2.txt 2>&1
cat $out/r4k_variants2.txt
