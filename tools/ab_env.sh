#!/bin/bash
# A/B of one environment switch inside ONE gpurun job: tools/ab_env.sh VAR=a VAR=b [steps]
set -e
A=$1; B=$2; steps=${3:-8}
mkdir -p gpurun_out
for round in 1 2; do
  env $A timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events > gpurun_out/abe_a_$round.json 2> gpurun_out/abe_a_$round.err
  env $B timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events > gpurun_out/abe_b_$round.json 2> gpurun_out/abe_b_$round.err
done
python - "$A" "$B" <<'PY'
import json, sys
for tag, name in (("a", sys.argv[1]), ("b", sys.argv[2])):
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/abe_{tag}_{r}.json").read().strip().splitlines()[-1])
        print(name, r, d["value"], d["ms_per_step"])
PY
