#!/bin/bash
# round 4, job I: masked data gradient on the two-board kernel (KA_CONV_PC2=3) against conv3x3_kernel's (2) in the step; then the GPU suite
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for round in 1 2; do
  for v in 2 3; do
    KA_CONV_PC2=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4i_bench_${v}_$round.json 2> $out/r4i_bench_${v}_$round.err || { tail -5 $out/r4i_bench_${v}_$round.err; exit 1; }
    python - $out/r4i_bench_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pc2", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), flush=True)
PY
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r4i_gpu_tests.txt 2>&1; rc=$?
tail -6 $out/r4i_gpu_tests.txt
exit $rc
