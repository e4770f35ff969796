#!/bin/bash
# round 4: the whole GPU suite on the current tree + default bench
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/r4full_tests.log 2>&1 || { tail -30 $out/r4full_tests.log; exit 1; }
tail -2 $out/r4full_tests.log
for r in 1 2; do timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130; done
