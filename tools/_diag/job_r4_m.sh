#!/bin/bash
# round 4, job M: the two-board conv at 128 channels (BASELINE configs[1]): parity against conv3x3_kernel, stand-alone times at B = 2048, the 6x128 line
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
CB_C=128 CB_B=2048 timeout -k 10 300 python tools/_diag/pc2_check.py > $out/r4m_pc2_c128.txt 2>&1 || { tail -20 $out/r4m_pc2_c128.txt; exit 1; }
grep "B=515\|B=2048\|vs fp32" $out/r4m_pc2_c128.txt | tail -16
for v in 0 3; do
  KA_CONV_PC2=$v timeout -k 10 300 python bench.py --workload 6x128 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 > $out/r4m_bench_6x128_pc2_$v.json 2> $out/r4m_bench_6x128_$v.err || { tail -5 $out/r4m_bench_6x128_$v.err; exit 1; }
  python - $out/r4m_bench_6x128_pc2_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("6x128 pc2", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("conv3x3_forward_launches_only"), d.get("wgrad_kernel"), flush=True)
PY
done
