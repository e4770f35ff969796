#!/bin/bash
# transformer evidence only (kernel stats + the two --pmc passes), as tools/profile_round.sh takes it
tag=r04
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tfstats -o t -- python3 $root/bench.py --workload transformer --no-cpu-baseline > $out/${tag}_transformer_bench_under_rocprof.json 2> $out/${tag}_tfstats.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_tffetch -o f -- python3 $root/bench.py --workload transformer --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_tffetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_tfwrite -o w -- python3 $root/bench.py --workload transformer --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_tfwrite.log 2>&1
cd $root
python3 tools/pmc_traffic.py $(find $out/${tag}_tffetch -name "*counter_collection.csv" | head -1) $(find $out/${tag}_tfwrite -name "*counter_collection.csv" | head -1) $out/${tag}_transformer_hbm_traffic.json "bench.py --workload transformer (d 256, 8 heads, 6 layers, minibatch 4096, bf16), 1 warm-up + 1 timed step"
python3 - $out/${tag}_transformer_hbm_traffic.json $out/${tag}_transformer_bench_under_rocprof.json <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
tot = sum(k["launches"] * k["hbm_bytes_per_launch"] for k in t["kernels"].values()) / 2        # two steps in the trace
ms = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["ms_per_step"]
t["step_summary"] = {"hbm_bytes_per_step": int(tot), "ms_per_step_under_rocprof_stats": ms, "achieved_TBps": tot / ms / 1e9, "frac_of_8TBps": tot / ms / 1e9 / 8.0}
json.dump(t, open(sys.argv[1], "w"), indent=1)
print("transformer step:", t["step_summary"])
PY
