#!/bin/bash
# round 4: pack kernel with unconditional reads, Adam in 16-byte pieces: parity, kernel stats of a short default bench run
set -e
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
KA_CHECK_ARGS=1 timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py tests/test_hip_ppo.py tests/test_hip_model.py tests/test_hip_transformer.py -x -q -m gpu > $out/small_tests.txt 2>&1 || { tail -30 $out/small_tests.txt; exit 1; }
tail -2 $out/small_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/small_stats -o s -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events > $out/small_bench_under_rocprof.json 2> $out/small_stats.err
cd $root
python3 - $(find $out/small_stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if any(k in r['Name'] for k in ('adam_kernel', 'pack_conv3x3', 'sqnorm', 'conv3x3_pc2_kernel<256, 5, false')):
        print(r['Name'].replace('(anonymous namespace)::','')[:80].ljust(80), r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
tail -1 $out/small_bench_under_rocprof.json | cut -c1-200
