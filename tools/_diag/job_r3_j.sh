#!/bin/bash
# round 3, job J: what the sixth row tile (square 80 + 15 zero rows) of the 3x3 convolutions costs -- five-tile build (libka_mt5.so,
# square 80 missing: wrong there, finite everywhere) against the product on random data, stand-alone launches (nothing feeds back)
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for v in base mt5; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib MFMA_ONE=fwd,fwd2,dgrad,dgradm MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -4
done
done
} > gpurun_out/r3_mt5.txt 2>&1
cat gpurun_out/r3_mt5.txt
