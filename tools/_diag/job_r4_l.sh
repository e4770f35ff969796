#!/bin/bash
# round 4, job L: GPU suite + default bench on the build with SLP off in conv3x3.hip; wgrad.hip with SLP off as a variant (stand-alone + step)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r4l_gpu_tests.txt 2>&1; rc=$?
tail -4 $out/r4l_gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for v in base wgnoslp; do
  lib=keisei_amd/libka_$v.so; [ $v = base ] && lib=keisei_amd/libkeisei_amd.so
  echo "== $v"; KEISEI_AMD_LIB=$PWD/$lib MFMA_ONE=fwd2,wgrad,wgradf MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py 2>&1 | tail -2
  KEISEI_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4l_bench_${v}_$rep.json 2> $out/r4l_bench_${v}_$rep.err || { tail -5 $out/r4l_bench_${v}_$rep.err; exit 1; }
  python - $out/r4l_bench_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("wgrad_kernel"), flush=True)
PY
done
done
