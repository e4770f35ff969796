#!/bin/bash
# round 4: in-kernel corner (KA_CONV_CORNER_IN=1): parity against the corner launch, stand-alone and in-step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "in_kernel_corner" > $out/r4cin_tests.log 2>&1 || { tail -30 $out/r4cin_tests.log; exit 1; }
tail -2 $out/r4cin_tests.log
for r in 1 2; do
  echo "== corner launch"; KA_CONV_CORNER_IN=0 MFMA_ONE=fwd,fwd2,dgrad,dgradm MFMA_ONE_TIME=1 MFMA_ONE_N=60 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep -E "fwd|dgrad"
  echo "== in-kernel corner"; KA_CONV_CORNER_IN=1 MFMA_ONE=fwd,fwd2,dgrad,dgradm MFMA_ONE_TIME=1 MFMA_ONE_N=60 timeout -k 10 200 python tools/mfma_one.py 2>&1 | grep -E "fwd|dgrad"
done > $out/r4cin_standalone.txt 2>&1
cat $out/r4cin_standalone.txt
run() { timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | cut -c1-130; }
for r in 1 2 3; do
  echo "corner launch"; KA_CONV_CORNER_IN=0 run
  echo "in-kernel corner (default: not the masked form)"; run
done > $out/r4cin_ab.txt
cat $out/r4cin_ab.txt
