#!/bin/bash
# round 4: LayerNorm backward with two row groups' loads up front: parity, kernel stats of the transformer workload
set -e
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
KA_CHECK_ARGS=1 timeout -k 10 600 python -m pytest tests/test_hip_transformer.py -x -q -m gpu > $out/ln_tests.txt 2>&1 || { tail -30 $out/ln_tests.txt; exit 1; }
tail -2 $out/ln_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ln_stats -o s -- python3 $root/bench.py --workload transformer --no-cpu-baseline > $out/ln_bench.json 2> $out/ln_stats.err
cd $root
python3 - $(find $out/ln_stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r['Name'].replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
tail -1 $out/ln_bench.json | cut -c1-200
