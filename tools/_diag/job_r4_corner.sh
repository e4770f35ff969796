#!/bin/bash
# round 4: what an in-kernel corner pass would at least cost: the forward two-board conv with the corner tile's weight stream + 64 MFMAs
# per wave appended after the last unit (ablation build libka_ctail.so, result discarded) against the shipped kernel; both runs still
# launch conv3x3_corner_kernel, so the difference is the in-kernel tail alone.  Stand-alone launches at B = 4096, alternating.
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for r in 1 2 3; do
  echo "== shipped"; MFMA_ONE=fwd,fwd2 MFMA_ONE_TIME=1 MFMA_ONE_N=60 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep fwd
  echo "== with the corner tail"; KEISEI_AMD_LIB=$PWD/keisei_amd/libka_ctail.so MFMA_ONE=fwd,fwd2 MFMA_ONE_TIME=1 MFMA_ONE_N=60 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep fwd
done > $out/r4_corner_tail.txt 2>&1
cat $out/r4_corner_tail.txt
