#!/bin/bash
# round 4, job O: the evidence on the shipped kernels -- held clock (stamps) and SQ / GRBM counters
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/held_clock.py r04_box$1 || exit 1
bash tools/mfma_counters.sh r04 > $out/r04_mfma_counters.log 2>&1 || { tail -5 $out/r04_mfma_counters.log; exit 1; }
tail -3 $out/r04_mfma_counters.log
