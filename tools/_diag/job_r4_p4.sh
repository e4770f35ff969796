#!/bin/bash
# round 4: half-width pieces in the gate-form boundary launch: parity, the step alternating
set -e
mkdir -p gpurun_out
KA_CHECK_ARGS=1 timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py tests/test_hip_fullsize.py tests/test_hip_ppo.py -x -q -m gpu > gpurun_out/p4_tests.txt 2>&1 || { tail -40 gpurun_out/p4_tests.txt; exit 1; }
tail -2 gpurun_out/p4_tests.txt
rm -f gpurun_out/p4_step_ab.txt
for round in 1 2 3; do
  for v in "KA_TAIL_GATE_P4=0" "KA_TAIL_GATE_P4=1"; do
    env $v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events > gpurun_out/p4_ab.json 2> gpurun_out/p4_ab.err || { tail -20 gpurun_out/p4_ab.err; exit 1; }
    python - "$v" $round <<'PY' | tee -a gpurun_out/p4_step_ab.txt
import json, sys
d = json.loads(open("gpurun_out/p4_ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["train_metrics"]["policy_loss"], d["train_metrics"]["value_loss"])
PY
  done
done
