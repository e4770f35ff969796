#!/bin/bash
# round 4, job G: border tiles in the two-board conv (KA_CONV_PC2_SKIP) -- parity, stand-alone times, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for v in 1 0; do
  echo "== KA_CONV_PC2_SKIP=$v"
  KA_CONV_PC2_SKIP=$v timeout -k 10 300 python tools/_diag/pc2_check.py > $out/r4g_pc2_check_skip$v.txt 2>&1 || { tail -20 $out/r4g_pc2_check_skip$v.txt; exit 1; }
  grep "B=4096\|vs fp32" $out/r4g_pc2_check_skip$v.txt | tail -10
done
for round in 1 2; do
  for v in 0 1; do
    KA_CONV_PC2_SKIP=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4g_bench_${v}_$round.json 2> $out/r4g_bench_${v}_$round.err || { tail -5 $out/r4g_bench_${v}_$round.err; exit 1; }
    python - $out/r4g_bench_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pc2_skip", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), {k: round(v, 4) for k, v in d["train_metrics"].items() if k in ("policy_loss", "value_loss")}, flush=True)
PY
  done
done
