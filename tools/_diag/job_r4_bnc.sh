#!/bin/bash
# BatchNorm coefficient kernels with all 64 partials of a sum in flight (this build) against batches of 16 (libka_old.so): parity, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "bn or stat or golden or mid_models or schedules" > $out/r4bnc_tests.log 2>&1 || { tail -30 $out/r4bnc_tests.log; exit 1; }
tail -1 $out/r4bnc_tests.log
bash tools/ab_bench.sh keisei_amd/libka_old.so 8 > $out/r4bnc_ab.txt 2>&1 || { tail -5 $out/r4bnc_ab.txt; exit 1; }
cat $out/r4bnc_ab.txt
