#!/bin/bash
# round 4, job Q: du-chain form of the block-boundary launch: parity, stand-alone time, step
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q > $out/r4q_tests.log 2>&1 || { tail -20 $out/r4q_tests.log; exit 1; }
tail -2 $out/r4q_tests.log
timeout -k 10 120 python tools/board_bench.py 2>&1 | grep -E "tail_bwd|block_dx" > $out/r4q_board.txt || exit 1
cat $out/r4q_board.txt
for r in 1 2; do
  KA_DX_TAIL=0 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
done > $out/r4q_ab.txt
cat $out/r4q_ab.txt
