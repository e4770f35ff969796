#!/bin/bash
# round 4: does the step time drift with the length of the timed region (8 / 30 / 100 steps after 3 warm-up steps)?
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for n in 8 30 100 8; do
  timeout -k 10 300 python bench.py --steps $n --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['ms_per_step'])"
done > $out/r4_sustain.txt
cat $out/r4_sustain.txt
