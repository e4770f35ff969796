#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_transformer.py tests/test_hip_kernels.py -m gpu -x -q > $out/r4tf3_tests.log 2>&1 || { tail -30 $out/r4tf3_tests.log; exit 1; }
tail -2 $out/r4tf3_tests.log
for r in 1 2; do timeout -k 10 300 python bench.py --workload transformer --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
bash tools/_diag/job_r4_tf2.sh
