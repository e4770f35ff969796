#!/bin/bash
# SQ counters of the transformer step's kernels, grouped by kernel and grid (bench.py --workload transformer, 2 steps)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/tfpmc -o c -- python3 $root/bench.py --workload transformer --steps 2 --warmup 1 --no-cpu-baseline > $out/tfpmc.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tfpmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""); k = k[:k.index("(")] if "(" in k else k
    key = (k[:34], r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[key] += 1
rows = []
for key, c in acc.items():
    L = max(n[key], 1); wc = max(c["SQ_WAVE_CYCLES"], 1)
    rows.append((c["GRBM_GUI_ACTIVE"] / 8, key, L, c))
for gpu_cyc, key, L, c in sorted(rows, reverse=True)[:16]:
    wc = max(c["SQ_WAVE_CYCLES"], 1)
    print(f"{key[0]:34s} grid {key[1]:>9s} x{L:3d}  gpu cycles/launch {gpu_cyc / L / 1e3:8.0f} k  pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(gpu_cyc * 1024, 1):5.2f}  wait {c['SQ_WAIT_ANY'] / wc:5.2f}  stall {c['SQ_WAIT_INST_ANY'] / wc:5.2f}  issue {c['SQ_ACTIVE_INST_ANY'] / wc:5.2f}  vmem/launch {c['SQ_INSTS_VMEM_RD'] / L / 1e6:6.2f} M")
PY
