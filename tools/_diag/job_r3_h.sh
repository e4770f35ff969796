#!/bin/bash
# round 3, job H: the K = 256 activation-stationary NT GEMM -- tests and the transformer step with / without it
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_transformer.py -x -q -m gpu > gpurun_out/r3_h_tests.txt 2>&1; tail -4 gpurun_out/r3_h_tests.txt
for cfg in "KA_TF_K256=0" "X=0" "KA_TF_K256=0" "X=0"; do
  env $cfg timeout -k 10 300 python bench.py --workload transformer --no-cpu-baseline > gpurun_out/r3_h.json 2>gpurun_out/r3_h.err
  python -c "import json;d=json.loads(open('gpurun_out/r3_h.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'])"
done
