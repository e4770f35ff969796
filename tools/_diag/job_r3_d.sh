#!/bin/bash
# round 3, job D: SQ counters of every MFMA kernel form (dgrad / wgrad had none), stand-alone times, staging-wave count in the step
mkdir -p gpurun_out
MFMA_ONE_TIME=1 MFMA_ONE_N=20 timeout -k 10 200 python tools/mfma_one.py > gpurun_out/r3_mfma_times.txt 2>&1; cat gpurun_out/r3_mfma_times.txt
bash tools/mfma_counters.sh r03 > gpurun_out/r3_mfma_counters.log 2>&1; tail -3 gpurun_out/r3_mfma_counters.log
for cfg in "X=0" "KA_CONV_P_NPW=2" "X=0" "KA_CONV_P_NPW=2" "KA_WGRAD_OVERLAP=0"; do
  env $cfg timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events > gpurun_out/r3_d.json 2>gpurun_out/r3_d.err
  python -c "import json;d=json.loads(open('gpurun_out/r3_d.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'])"
done
