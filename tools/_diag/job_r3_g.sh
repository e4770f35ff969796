#!/bin/bash
# round 3, job G: VERDICT r2 item 1(d) -- v_mfma_f32_32x32x16_bf16 against 16x16x32: register-resident burn of both shapes, and the
# producer/consumer conv with its MFMAs swapped for the same pipe cycles of the larger shape (libka_m32.so: WRONG results, timing only)
mkdir -p gpurun_out
{
timeout -k 10 200 python tools/_diag/mfma_shapes.py 2>&1 | grep -v warning
for rep in 1 2 3; do
for v in base m32; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib MFMA_ONE=fwd,fwd2,dgrad MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -3
done
done
} > gpurun_out/r3_mfma32.txt 2>&1
cat gpurun_out/r3_mfma32.txt
