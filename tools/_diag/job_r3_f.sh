#!/bin/bash
# round 3, job F: block_dx + tail_bwd in one launch -- bit-identity tests, stand-alone times, A/B in the step
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "block_dx_tail or tail_bwd" > gpurun_out/r3_f_tests.txt 2>&1; tail -3 gpurun_out/r3_f_tests.txt
timeout -k 10 200 python tools/board_bench.py > gpurun_out/r3_board_bench.txt 2>&1; cat gpurun_out/r3_board_bench.txt
for cfg in "KA_DX_TAIL=0" "X=0" "KA_DX_TAIL=0" "X=0"; do
  env $cfg timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events > gpurun_out/r3_f.json 2>gpurun_out/r3_f.err
  python -c "import json;d=json.loads(open('gpurun_out/r3_f.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'])"
done
timeout -k 10 900 python -m pytest tests/test_hip_model.py tests/test_hip_ppo.py tests/test_hip_fullsize.py -x -q -m gpu > gpurun_out/r3_f_tests2.txt 2>&1; tail -3 gpurun_out/r3_f_tests2.txt
