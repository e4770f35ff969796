#!/bin/bash
# round 4, job X3: GPU-busy time of a 6x128 step (kernel trace) against its wall time
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r4x3_stats -o s -- python3 $root/bench.py --workload 6x128 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --no-secondary --no-kernel-events > $out/r4x3_bench.json 2> $out/r4x3.err || exit 1
cd $root
tail -1 $out/r4x3_bench.json | cut -c1-160
python3 - $(find $out/r4x3_stats -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time {tot / 45 / 1e6:.3f} ms per step over 45 steps, {calls / 45:.0f} launches per step")
for r in rows[:14]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), f"{float(r['TotalDurationNs']) / 45 / 1e3:8.1f} us/step", f"{float(r['AverageNs']) / 1e3:7.1f} us")
PY
