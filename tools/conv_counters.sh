#!/bin/bash
# SQ / GRBM counters of the stand-alone conv3x3 launch loop (tools/conv_one.py): matrix-pipe busy share, wave wait shares,
# effective clock.  One --pmc pass (no tracing domains besides the kernel trace).   tools/conv_counters.sh <tag>
tag=${1:-r02}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_convpmc -o c -- python3 $root/tools/conv_one.py > $out/${tag}_convpmc.log 2>&1
cd $root
python3 - "$out/${tag}_convpmc" "$out/${tag}_conv_sq_counters.json" <<'PY'
import csv, glob, json, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "conv3x3_kernel" not in k and "conv3x3_pc_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
out = {}
for k, c in acc.items():
    L = max(n[k], 1)
    busy_cu = c.get("SQ_BUSY_CU_CYCLES", 0) / L
    out[k[:60]] = {"launches": L, "per_launch": {m: round(v / L, 1) for m, v in c.items()},
                   "mfma_busy_share_of_cu_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CU_CYCLES", 1), 1), 4),
                   "matrix_pipe_busy_fraction (MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 4),
                   "wave_cycle_shares": {m: round(c.get(m, 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 4) for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}}
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/conv_one.py (B = 4096, C = 256, bf16, plain forward conv, 5 launches)",
           "note": "raw counter sums over all XCDs / SEs as rocprofv3 reports them; shares are ratios of like counters", "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
PY
