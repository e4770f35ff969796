#!/bin/bash
# round 4, job P: backward tail with batched FC weight loads (this build) against the previous build, plus the second box's held clock
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "tail or block or fused or parity or grad" > $out/r4p_tests.log 2>&1 || { tail -20 $out/r4p_tests.log; exit 1; }
tail -2 $out/r4p_tests.log
for r in 1 2; do
  echo "== old"; KEISEI_AMD_LIB=keisei_amd/libka_oldtail.so timeout -k 10 120 python tools/board_bench.py 2>&1 | grep -E "tail_bwd|block_dx"
  echo "== new"; timeout -k 10 120 python tools/board_bench.py 2>&1 | grep -E "tail_bwd|block_dx"
done > $out/r4p_board.txt 2>&1
cat $out/r4p_board.txt
bash tools/ab_bench.sh keisei_amd/libka_oldtail.so 8 > $out/r4p_ab.txt 2>&1 || { tail -5 $out/r4p_ab.txt; exit 1; }
cat $out/r4p_ab.txt
timeout -k 10 300 python tools/held_clock.py r04_box2 > $out/r4p_clock.log 2>&1 || { tail -5 $out/r4p_clock.log; exit 1; }
tail -3 $out/r4p_clock.log
