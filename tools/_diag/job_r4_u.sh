#!/bin/bash
# round 4, job U: wgrad_flat with the lean staging: staggered (waves 4-7 stage after their k-steps) against all-waves-stage-first
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/wgrad_lean_check.py > $out/r4u_check.txt 2>&1 || { tail -5 $out/r4u_check.txt; exit 1; }
grep -v amdgpu.ids $out/r4u_check.txt | tail -6
KEISEI_AMD_LIB=keisei_amd/libka_wgtl.so timeout -k 10 200 python tools/_diag/wgrad_tl.py > $out/r4u_wgrad_tl.txt 2>&1 || { tail -5 $out/r4u_wgrad_tl.txt; exit 1; }
cat $out/r4u_wgrad_tl.txt
for r in 1 2; do
  KA_WGRAD_STAG=0 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
done > $out/r4u_ab.txt
cat $out/r4u_ab.txt
