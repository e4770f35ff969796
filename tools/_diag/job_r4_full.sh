#!/bin/bash
# round 4: the whole GPU suite on the current tree, then the two-stream backward (KA_WGRAD_OVERLAP=1) against the default
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/r4full_tests.log 2>&1 || { tail -30 $out/r4full_tests.log; exit 1; }
tail -2 $out/r4full_tests.log
run() { timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130; }
for r in 1 2; do
  echo "default"; run
  echo "KA_WGRAD_OVERLAP=1"; KA_WGRAD_OVERLAP=1 run
done > $out/r4full_overlap_ab.txt
cat $out/r4full_overlap_ab.txt
