#!/bin/bash
# round 3, job E: FC chain on the main stream vs side stream (A/B), SyncBN exposure in the dry run, DDP + model tests
mkdir -p gpurun_out
for cfg in "KA_FC_SIDE=1" "X=0" "KA_FC_SIDE=1" "X=0"; do
  env $cfg timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events > gpurun_out/r3_e.json 2>gpurun_out/r3_e.err
  python -c "import json;d=json.loads(open('gpurun_out/r3_e.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'])"
done
timeout -k 10 300 python bench.py --dist-dry-run > gpurun_out/r3_dry.json 2>gpurun_out/r3_dry.err; tail -1 gpurun_out/r3_dry.json
timeout -k 10 900 python -m pytest tests/test_hip_ddp.py tests/test_hip_model.py tests/test_hip_ppo.py -x -q -m gpu > gpurun_out/r3_e_tests.txt 2>&1; tail -5 gpurun_out/r3_e_tests.txt
