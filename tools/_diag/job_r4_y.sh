#!/bin/bash
# round 4, job Y: the global-pool FC chains forked beside the statistics kernels (KA_FC_SIDE=2 forward, KA_FC_BWD_SIDE=1 backward)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_model.py -m gpu -x -q -k "schedules" > $out/r4y_tests.log 2>&1 || { tail -20 $out/r4y_tests.log; exit 1; }
tail -2 $out/r4y_tests.log
run() { timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130; }
for r in 1 2; do
  echo "default"; run
  echo "KA_FC_BWD_SIDE=1"; KA_FC_BWD_SIDE=1 run
  echo "KA_FC_SIDE=2"; KA_FC_SIDE=2 run
  echo "both"; KA_FC_SIDE=2 KA_FC_BWD_SIDE=1 run
done > $out/r4y_ab.txt
cat $out/r4y_ab.txt
