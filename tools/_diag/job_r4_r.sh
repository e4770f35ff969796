#!/bin/bash
# round 4, job R: backward tail with its per-channel FC operands requested before the board's loads (this build) vs job Q's build
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "tail or block_dx" > $out/r4r_tests.log 2>&1 || { tail -20 $out/r4r_tests.log; exit 1; }
tail -2 $out/r4r_tests.log
for r in 1 2; do
  echo "== job Q build"; KEISEI_AMD_LIB=keisei_amd/libka_q.so timeout -k 10 120 python tools/board_bench.py 2>&1 | grep -E "tail_bwd_fused|block_dx_tail"
  echo "== this build"; timeout -k 10 120 python tools/board_bench.py 2>&1 | grep -E "tail_bwd_fused|block_dx_tail"
done > $out/r4r_board.txt 2>&1
cat $out/r4r_board.txt
bash tools/ab_bench.sh keisei_amd/libka_q.so 8 > $out/r4r_ab.txt 2>&1 || { tail -5 $out/r4r_ab.txt; exit 1; }
cat $out/r4r_ab.txt
