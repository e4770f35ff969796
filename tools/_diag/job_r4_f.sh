#!/bin/bash
# round 4, job F: lean weight gradient -- bit identity, stand-alone times, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/wgrad_lean_check.py > $out/r4f_wgrad_lean.txt 2>&1 || { tail -20 $out/r4f_wgrad_lean.txt; exit 1; }
cat $out/r4f_wgrad_lean.txt
for round in 1 2; do
  for v in 0 1; do
    KA_WGRAD_LEAN=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4f_bench_${v}_$round.json 2> $out/r4f_bench_${v}_$round.err || { tail -5 $out/r4f_bench_${v}_$round.err; exit 1; }
    python - $out/r4f_bench_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("wgrad_lean", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("wgrad_kernel"), flush=True)
PY
  done
done
