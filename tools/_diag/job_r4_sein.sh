#!/bin/bash
# round 4: the squeeze-excite FC chain inside the forward-tail launch (KA_SE_IN_TAIL): parity, then the step alternating
set -e
mkdir -p gpurun_out
KA_CHECK_ARGS=1 timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -x -q -m gpu -k "tail_forward or se_chain_inside" > gpurun_out/sein_tests.txt 2>&1 || { tail -40 gpurun_out/sein_tests.txt; exit 1; }
tail -2 gpurun_out/sein_tests.txt
rm -f gpurun_out/sein_step_ab.txt
for round in 1 2 3; do
  for v in "KA_SE_IN_TAIL=0" "KA_SE_IN_TAIL=1"; do
    env $v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events > gpurun_out/sein_ab.json 2> gpurun_out/sein_ab.err || { tail -20 gpurun_out/sein_ab.err; exit 1; }
    python - "$v" $round <<'PY' | tee -a gpurun_out/sein_step_ab.txt
import json, sys
d = json.loads(open("gpurun_out/sein_ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["train_metrics"]["policy_loss"], d["train_metrics"]["value_loss"])
PY
  done
done
