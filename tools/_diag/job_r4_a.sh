#!/bin/bash
# round 4, job A: this round's starting point on one box -- the default bench line, stand-alone times of every MFMA form, and the
# vector-memory / LDS path counters of the tower convolutions (is the per-CU load path what the matrix pipe waits for?)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 > $out/r4a_bench.json 2> $out/r4a_bench.err || exit 1
tail -1 $out/r4a_bench.json | cut -c1-400
MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py > $out/r4a_standalone.txt 2>&1 || exit 1
cat $out/r4a_standalone.txt
cd /tmp && export TMPDIR=/tmp
export MFMA_ONE=fwd,fwd2,dgrad,dgradm
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r4a_pmc1 -o c -- python3 $root/tools/mfma_one.py > $out/r4a_pmc1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r4a_pmc2 -o c -- python3 $root/tools/mfma_one.py > $out/r4a_pmc2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r4a_pmc3 -o c -- python3 $root/tools/mfma_one.py > $out/r4a_pmc3.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r4a_pmc4 -o c -- python3 $root/tools/mfma_one.py > $out/r4a_pmc4.log 2>&1
rc=$?
cd $root
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in sorted(glob.glob(out + "/r4a_pmc[0-9]")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "conv3x3" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k[:k.index("(")] if "(" in k else k
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
res = {k: {m: round(v / max(n[k][m], 1), 1) for m, v in c.items()} for k, c in acc.items()}
json.dump(res, open(out + "/r4a_vmem_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
exit $rc
