root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cstats -o s -- python3 $root/bench.py --no-cpu-baseline --no-fp32 --steps 6 --warmup 2 > $out/cstats.json 2> $out/cstats.err
cd $root
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/cstats/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
