#!/bin/bash
# round 4, job D: the whole GPU suite on the build with the two-board conv as the default, then the default bench line
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r4d_gpu_tests.txt 2>&1; rc=$?
tail -15 $out/r4d_gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r4d_bench.json 2> $out/r4d_bench.err || { tail -5 $out/r4d_bench.err; exit 1; }
python - $out/r4d_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("conv3x3_forward_launches_only"), d.get("wgrad_kernel"))
print(json.dumps(d.get("secondary"), indent=1)[:2500])
print(d.get("fp32_mode"))
PY
