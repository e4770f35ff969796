#!/bin/bash
# round 3, job C: what paces the producer/consumer conv's MFMA loop -- ablation builds (wrong results, timing only)
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in base nw na nwa; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib CB_QUICK=1 timeout -k 10 200 python tools/conv_bench.py conv 2>&1 | tail -4 | head -2
done
for abl in 1 2 3; do
  echo "== diag KA_CONV_P_ABL=$abl"; KA_CONV_P_ABL=$abl KEISEI_AMD_LIB=$PWD/keisei_amd/libka_diag.so CB_QUICK=1 timeout -k 10 200 python tools/conv_bench.py conv 2>&1 | tail -4 | head -2
done
for abl in 0 1 2 3; do
  echo "== nwa KA_CONV_P_ABL=$abl"; KA_CONV_P_ABL=$abl KEISEI_AMD_LIB=$PWD/keisei_amd/libka_nwa.so CB_QUICK=1 timeout -k 10 200 python tools/conv_bench.py conv 2>&1 | tail -4 | head -2
done
done
} > gpurun_out/r3_pc_ablations.txt 2>&1
cat gpurun_out/r3_pc_ablations.txt
