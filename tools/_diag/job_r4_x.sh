#!/bin/bash
# round 4, job X: host enqueue time per step against the step time, headline and 6x128
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
KA_HOST_TIMING=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4x_host_40x256.json 2> $out/r4x_host_40x256.err || exit 1
grep "host enqueue" $out/r4x_host_40x256.err; tail -1 $out/r4x_host_40x256.json | cut -c1-200
KA_HOST_TIMING=1 timeout -k 10 300 python bench.py --workload 6x128 --steps 30 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4x_host_6x128.json 2> $out/r4x_host_6x128.err || exit 1
grep "host enqueue" $out/r4x_host_6x128.err; tail -1 $out/r4x_host_6x128.json | cut -c1-200
