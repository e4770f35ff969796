#!/bin/bash
# SQ / GRBM counters of every MFMA kernel form of the step (tools/mfma_one.py), stand-alone launches at B = 4096, C = 256:
# matrix-pipe busy share, wave wait shares, LDS bank conflicts.  Two --pmc passes (8 SQ slots each), kernel trace only.
#   tools/mfma_counters.sh <tag>     -> gpurun_out/<tag>_mfma_sq_counters.json
tag=${1:-r03}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_mfmapmc1 -o c -- python3 $root/tools/mfma_one.py > $out/${tag}_mfmapmc1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_mfmapmc2 -o c -- python3 $root/tools/mfma_one.py > $out/${tag}_mfmapmc2.log 2>&1
rc=$?
cd $root
python3 - "$out/${tag}_mfmapmc1" "$out/${tag}_mfmapmc2" "$out/${tag}_mfma_sq_counters.json" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in sys.argv[1:3]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("conv3x3", "wgrad_kernel", "wgrad_flat", "conv_b", "wgrad2")): continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k[:k.index("(")] if "(" in k else k
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {}
for k, c in acc.items():
    per = {m: v / max(n[k][m], 1) for m, v in c.items()}
    wc = max(per.get("SQ_WAVE_CYCLES", 1), 1)
    out[k] = {"launches": n[k]["SQ_WAVE_CYCLES"] // 2 if "SQ_LDS_IDX_ACTIVE" in per and "SQ_WAIT_ANY" in per else n[k]["SQ_WAVE_CYCLES"],
              "per_launch": {m: round(v, 1) for m, v in per.items()},
              "matrix_pipe_busy_fraction": round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(per.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 4),
              "gpu_cycles_per_launch": round(per.get("GRBM_GUI_ACTIVE", 0) / 8, 0),
              "wave_cycle_shares": {m: round(per.get(m, 0) / wc, 4) for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")},
              "lds_bank_conflict_share_of_lds_cycles": round(per.get("SQ_LDS_BANK_CONFLICT", 0) / max(per.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4)}
json.dump({"command": "rocprofv3 --pmc <8 SQ/GRBM counters> --kernel-trace -- python3 tools/mfma_one.py, two passes (B = 4096, C = 256, bf16, 7 launches per form)",
           "note": "raw counter sums over all XCDs / SEs as rocprofv3 reports them, averaged per launch; SQ_VALU_MFMA_BUSY_CYCLES = 16 x MFMAs for v_mfma_f32_16x16x32_bf16; matrix_pipe_busy_fraction = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
exit $rc
