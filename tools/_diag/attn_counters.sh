#!/bin/bash
# SQ counters of the attention kernels (tools/attn_bench.py), one --pmc pass
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/attnpmc -o c -- python3 $root/tools/attn_bench.py > $out/attnpmc.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/attnpmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attention" not in k: continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", ""); k = k[:k.index("(")]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    L = n[k]; wc = c["SQ_WAVE_CYCLES"]
    print(k, "launches", L, {m: round(v / L) for m, v in c.items()}, "shares", {m: round(c[m] / wc, 3) for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")})
PY
