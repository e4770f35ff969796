#!/bin/bash
# round 4, job X2: host enqueue time per step after the launch-plan fast path in _lib.call
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for r in 1 2; do
KA_HOST_TIMING=1 timeout -k 10 300 python bench.py --workload 6x128 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4x2_host_6x128.json 2> $out/r4x2_host_6x128.err || exit 1
grep "host enqueue" $out/r4x2_host_6x128.err; tail -1 $out/r4x2_host_6x128.json | cut -c1-160
done
KA_HOST_TIMING=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4x2_host_40x256.json 2> $out/r4x2_host_40x256.err || exit 1
grep "host enqueue" $out/r4x2_host_40x256.err; tail -1 $out/r4x2_host_40x256.json | cut -c1-160
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r4x2_tests.log 2>&1 || { tail -20 $out/r4x2_tests.log; exit 1; }
tail -2 $out/r4x2_tests.log
