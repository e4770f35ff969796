#!/bin/bash
# round 4, job V: wgrad_flat k-steps as one software pipeline with the staging inside it
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/wgrad_lean_check.py > $out/r4v_check.txt 2>&1 || { tail -5 $out/r4v_check.txt; exit 1; }
grep -v amdgpu.ids $out/r4v_check.txt | tail -10
KEISEI_AMD_LIB=keisei_amd/libka_wgtl.so timeout -k 10 200 python tools/_diag/wgrad_tl.py > $out/r4v_wgrad_tl.txt 2>&1 || { tail -5 $out/r4v_wgrad_tl.txt; exit 1; }
cat $out/r4v_wgrad_tl.txt
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "wgrad" > $out/r4v_tests.log 2>&1 || { tail -20 $out/r4v_tests.log; exit 1; }
tail -2 $out/r4v_tests.log
bash tools/ab_bench.sh keisei_amd/libka_old.so 8 > $out/r4v_ab.txt 2>&1 || { tail -5 $out/r4v_ab.txt; exit 1; }
cat $out/r4v_ab.txt
