#!/bin/bash
# round 3, job I: the policy-layer NT GEMMs under the 8 x 8 super-tile XCD map: durations and L2-miss traffic by launch shape
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
export KA_TF_MAP2D=$v
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/map2d_$v -o f -- python3 $root/bench.py --workload transformer --steps 1 --warmup 1 --no-cpu-baseline > $out/map2d_$v.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1):
    f = glob.glob(f"gpurun_out/map2d_{v}/*counter_collection.csv")[0]
    t = glob.glob(f"gpurun_out/map2d_{v}/*kernel_trace.csv")[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t)):
        if "gemm_nt_bf16" in r["Kernel_Name"]:
            dur[r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_bf16" in r["Kernel_Name"]: agg[r["Grid_Size"]].append(float(r["Counter_Value"]))
    print("KA_TF_MAP2D =", v)
    for g in sorted(agg, key=lambda k: -sum(agg[k])):
        print(f"  grid {g:>9s} x{len(agg[g]):3d}  L2-miss reads {2 * 1024 * sum(agg[g]) / len(agg[g]) / 1e6:9.1f} MB per launch   durations (us, under --pmc) {[round(x) for x in dur.get(g, [])][:4]}")
PY
