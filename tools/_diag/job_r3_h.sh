#!/bin/bash
# round 3, job H: the producer/consumer data-gradient forms (KA_CONV_P=2: plain epilogue, 3: + masked epilogue) under the one-stream backward
mkdir -p gpurun_out
{
for p in 1 3; do
  echo "== stand-alone KA_CONV_P=$p"; KA_CONV_P=$p MFMA_ONE=dgrad,dgradm MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -2
done
for rep in 1 2; do
for p in 1 2 3; do
  echo "== bench KA_CONV_P=$p"; KA_CONV_P=$p timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
} > gpurun_out/r3_pc_dgrad.txt 2>&1
cat gpurun_out/r3_pc_dgrad.txt
