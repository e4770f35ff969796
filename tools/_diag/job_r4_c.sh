#!/bin/bash
# round 4, job C: two boards per weight fragment (conv3x3_pc2_kernel) -- parity against the one-board form, stand-alone times, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/pc2_check.py > $out/r4c_pc2_check.txt 2>&1 || { tail -20 $out/r4c_pc2_check.txt; exit 1; }
cat $out/r4c_pc2_check.txt
for round in 1 2; do
  for v in 0 2 3; do
    KA_CONV_PC2=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4c_bench_pc2_${v}_$round.json 2> $out/r4c_bench_pc2_${v}_$round.err || { tail -5 $out/r4c_bench_pc2_${v}_$round.err; exit 1; }
    python - $out/r4c_bench_pc2_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pc2", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), {k: round(v, 4) for k, v in d["train_metrics"].items() if k in ("policy_loss", "value_loss")}, flush=True)
PY
  done
done
