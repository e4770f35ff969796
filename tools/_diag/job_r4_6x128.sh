#!/bin/bash
# kernel stats of the 6x128 workload (BASELINE configs[1])
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s6_stats -o s -- python3 $root/bench.py --workload 6x128 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events > $out/s6_bench.json 2> $out/s6_stats.err
cd $root
python3 - $(find $out/s6_stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(r['Name'].replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
tail -1 $out/s6_bench.json | cut -c1-250
