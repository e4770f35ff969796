#!/bin/bash
# round 4, job J: staggered two-board conv (KA_CONV_PC2_STAG=1): parity, stand-alone times, step A/B
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for v in 1 0; do
  echo "== KA_CONV_PC2_STAG=$v"
  KA_CONV_PC2_STAG=$v timeout -k 10 300 python tools/_diag/pc2_check.py > $out/r4j_pc2_check_stag$v.txt 2>&1 || { tail -20 $out/r4j_pc2_check_stag$v.txt; exit 1; }
  grep "B=515\|B=4096" $out/r4j_pc2_check_stag$v.txt | tail -12
done
for round in 1 2; do
  for v in 0 1; do
    KA_CONV_PC2_STAG=$v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4j_bench_${v}_$round.json 2> $out/r4j_bench_${v}_$round.err || { tail -5 $out/r4j_bench_${v}_$round.err; exit 1; }
    python - $out/r4j_bench_${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pc2_stag", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), flush=True)
PY
  done
done
