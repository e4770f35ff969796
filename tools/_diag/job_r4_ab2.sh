#!/bin/bash
# round 4: the weight-gradient launches INSIDE the step, previous staging (libka_oldstage.so) against the shipped one, alternating on one box
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for r in 1 2 3; do
  for v in oldstage shipped; do
    lib=keisei_amd/libkeisei_amd.so; [ $v = oldstage ] && lib=keisei_amd/libka_oldstage.so
    KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | tail -1 > $out/ab2_$v.json
    python3 - $v $out/ab2_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())

print(sys.argv[1], d["value"], d["ms_per_step"], "conv", d["roofline"]["avg_launch_ms"], "wgrad", d.get("wgrad_kernel"))
PY
  done
done > $out/r4_ab2.txt
cat $out/r4_ab2.txt
