#!/bin/bash
# round 4, job E: staggered weight gradient -- bit identity, stand-alone times, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/wgrad_stag_check.py > $out/r4e_wgrad_stag.txt 2>&1 || { tail -20 $out/r4e_wgrad_stag.txt; exit 1; }
cat $out/r4e_wgrad_stag.txt
for round in 1 2; do
  for v in 0 1; do
    KA_WGRAD_STAG=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4e_bench_${v}_$round.json 2> $out/r4e_bench_${v}_$round.err || { tail -5 $out/r4e_bench_${v}_$round.err; exit 1; }
    python - $out/r4e_bench_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("wgrad_stag", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("wgrad_kernel"), flush=True)
PY
  done
done
