#!/bin/bash
# round 4: BatchNorm statistics as one launch (stage-1 reduce + coefficients by the last-arriving workgroup): parity, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -x -q -k "bn or schedules or stat" > $out/r4bn_tests.log 2>&1 || { tail -30 $out/r4bn_tests.log; exit 1; }
tail -2 $out/r4bn_tests.log
run() { timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130; }
for r in 1 2 3; do
  echo "two launches"; KA_BN_ONE_LAUNCH=0 run
  echo "one launch"; run
done > $out/r4bn_ab.txt
cat $out/r4bn_ab.txt
KA_BN_ONE_LAUNCH=0 timeout -k 10 200 python bench.py --workload 6x128 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
timeout -k 10 200 python bench.py --workload 6x128 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130
