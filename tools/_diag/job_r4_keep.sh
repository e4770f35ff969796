#!/bin/bash
# round 4: conv2's transformed input kept by the forward (ka_conv3x3_fwd_keep) so that its weight gradient is the plain form: parity, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py tests/test_hip_fullsize.py -m gpu -x -q > $out/r4keep_tests.log 2>&1 || { tail -30 $out/r4keep_tests.log; exit 1; }
tail -2 $out/r4keep_tests.log
run() { timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'conv', d['roofline']['avg_launch_ms'], 'fwd', d['conv3x3_forward_launches_only']['avg_launch_ms'], 'wgrad', d['wgrad_kernel']['avg_launch_ms'])"; }
for r in 1 2 3; do
  echo "recompute in wgrad (KA_KEEP_X2=0)"; KA_KEEP_X2=0 run
  echo "kept by the forward"; run
done > $out/r4keep_ab.txt
cat $out/r4keep_ab.txt
