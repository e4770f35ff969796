#!/bin/bash
# round 4, job H: weights through buffer loads (scalar step offsets) + the running-index fragment ring in the two-board conv: parity, times, step
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/pc2_check.py > $out/r4h_pc2_check.txt 2>&1 || { tail -20 $out/r4h_pc2_check.txt; exit 1; }
grep "B=515\|B=4096\|vs fp32" $out/r4h_pc2_check.txt | tail -18
for round in 1 2; do
    timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4h_bench_$round.json 2> $out/r4h_bench_$round.err || { tail -5 $out/r4h_bench_$round.err; exit 1; }
    python - $out/r4h_bench_$round.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), d.get("wgrad_kernel"), flush=True)
PY
done
