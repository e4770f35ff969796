#!/bin/bash
# round 4: batched loads in the gate-form boundary launch and in block_tail_fwd16: parity, stand-alone, step A/B
set -e
mkdir -p gpurun_out
export KA_CHECK_ARGS=1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu > gpurun_out/gate2_tests.txt 2>&1 || { tail -30 gpurun_out/gate2_tests.txt; exit 1; }
tail -3 gpurun_out/gate2_tests.txt
timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py tests/test_hip_ppo.py -x -q -m gpu > gpurun_out/gate2_tests_model.txt 2>&1 || { tail -30 gpurun_out/gate2_tests_model.txt; exit 1; }
tail -3 gpurun_out/gate2_tests_model.txt
unset KA_CHECK_ARGS
timeout -k 10 200 python tools/board_bench.py > gpurun_out/gate2_board_bench.txt 2>&1
tail -9 gpurun_out/gate2_board_bench.txt
rm -f gpurun_out/gate2_step_ab.txt
for round in 1 2; do
  for v in "KA_TAIL_GATE=0" "KA_TAIL_GATE=1" "KA_TAIL_FWD_KB=2" "KA_TAIL_FWD_KB=6"; do
    env $v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events --no-secondary > gpurun_out/gate_ab.json 2> gpurun_out/gate_ab.err || { tail -20 gpurun_out/gate_ab.err; exit 1; }
    python - "$v" $round <<'PY' | tee -a gpurun_out/gate2_step_ab.txt
import json, sys
d = json.loads(open("gpurun_out/gate_ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["train_metrics"]["policy_loss"], d["train_metrics"]["value_loss"], d["train_metrics"]["gradient_norm"])
PY
  done
done
