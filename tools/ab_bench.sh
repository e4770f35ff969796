#!/bin/bash
# A/B of two builds of the library inside ONE gpurun job (box-to-box spread is +-2 %): tools/ab_bench.sh <other.so> [steps]
set -e
other=$1; steps=${2:-8}
mkdir -p gpurun_out
for round in 1 2; do
  KEISEI_AMD_LIB=$other timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-fp32 > gpurun_out/ab_other_$round.json 2> gpurun_out/ab_other_$round.err
  timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-fp32 > gpurun_out/ab_this_$round.json 2> gpurun_out/ab_this_$round.err
done
python - <<'PY'
import json
for tag in ("other", "this"):
    for r in (1, 2):
        d = json.loads(open(f"gpurun_out/ab_{tag}_{r}.json").read().strip().splitlines()[-1])
        print(tag, r, d["value"], d["ms_per_step"])
PY
