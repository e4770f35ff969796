#!/bin/bash
# round 3, job A: co-residency diagnostic, the weight-gradient forms against each other alone and in the step
mkdir -p gpurun_out
timeout -k 10 200 python tools/_diag/coresidency.py 64 96 104 124 > gpurun_out/r3_cores.txt 2>&1; tail -6 gpurun_out/r3_cores.txt
for V in 2 3 4; do
  WG_AB_V=$V timeout -k 10 300 python tools/wgrad_ab.py > gpurun_out/r3_wgrad_ab$V.txt 2>&1; echo "== V=$V"; grep "B=4096" gpurun_out/r3_wgrad_ab$V.txt | cut -c1-230; grep -c "bit-identical=True" gpurun_out/r3_wgrad_ab$V.txt
done
for v in 1 2 3 4 1 2 3 4; do
  KA_WGRAD_V=$v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 > gpurun_out/r3_ab_v$v.json 2>gpurun_out/r3_ab_v$v.err
  python -c "import json;d=json.loads(open('gpurun_out/r3_ab_v$v.json').read().strip().splitlines()[-1]);print('V=$v',d['value'],d['ms_per_step'],d.get('wgrad_kernel'),d['roofline']['avg_launch_ms'])"
done
