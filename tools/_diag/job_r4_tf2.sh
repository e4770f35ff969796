#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r4tf2_stats -o t -- python3 $root/bench.py --workload transformer --steps 10 --warmup 3 --no-cpu-baseline > $out/r4tf2_bench.json 2> $out/r4tf2.err || exit 1
cd $root
python3 - $(find $out/r4tf2_stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in ("weights16", "cast_pad", "transpose_pad", "reduce_slabs", "adam")):
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), f"{float(r['TotalDurationNs']) / 13 / 1e3:9.1f} us/step", f"{float(r['AverageNs']) / 1e3:8.1f} us")
PY
