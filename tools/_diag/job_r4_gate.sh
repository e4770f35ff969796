#!/bin/bash
# round 4, dz-less chain launch: parity tests, stand-alone timings, step A/B (KA_TAIL_GATE=0 / 1 / 1 at six waves per SIMD)
set -e
mkdir -p gpurun_out
export KA_CHECK_ARGS=1
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gated or block_dx_tail or two_board or dgrad" > gpurun_out/gate_tests.txt 2>&1 || { tail -30 gpurun_out/gate_tests.txt; exit 1; }
tail -3 gpurun_out/gate_tests.txt
timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py tests/test_hip_ppo.py -x -q -m gpu > gpurun_out/gate_tests_model.txt 2>&1 || { tail -30 gpurun_out/gate_tests_model.txt; exit 1; }
tail -3 gpurun_out/gate_tests_model.txt
unset KA_CHECK_ARGS
timeout -k 10 200 python tools/board_bench.py > gpurun_out/gate_board_bench.txt 2>&1
tail -5 gpurun_out/gate_board_bench.txt
CB_QUICK=1 timeout -k 10 200 python tools/conv_bench.py conv > gpurun_out/gate_conv_bench.txt 2>&1
cat gpurun_out/gate_conv_bench.txt
for round in 1 2; do
  for v in "KA_TAIL_GATE=0" "KA_TAIL_GATE=1" "KA_TAIL_GATE_WPE=6"; do
    env $v timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events --no-secondary > gpurun_out/gate_ab.json 2> gpurun_out/gate_ab.err || { tail -20 gpurun_out/gate_ab.err; exit 1; }
    python - "$v" $round <<'PY' | tee -a gpurun_out/gate_step_ab.txt
import json, sys
d = json.loads(open("gpurun_out/gate_ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["train_metrics"]["policy_loss"], d["train_metrics"]["value_loss"], d["train_metrics"]["gradient_norm"])
PY
  done
done
