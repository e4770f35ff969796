#!/bin/bash
# round 4: transformer: the bf16 weight cache in one launch + paired slab reduce: parity, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_transformer.py -m gpu -x -q > $out/r4tf_tests.log 2>&1 || { tail -30 $out/r4tf_tests.log; exit 1; }
tail -2 $out/r4tf_tests.log
run() { timeout -k 10 300 python bench.py --workload transformer --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; }
for r in 1 2; do
  echo "per-layer copies"; KA_TF_W16_MULTI=0 run
  echo "one launch"; run
done > $out/r4tf_ab.txt
cat $out/r4tf_ab.txt
bash tools/_diag/job_r4_tf2.sh
