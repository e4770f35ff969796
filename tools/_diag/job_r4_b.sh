#!/bin/bash
# round 4, job B: the staggered producer/consumer schedule (KA_CONV_P_STAG=1) -- bit identity, stand-alone times, the step A/B;
# then the vector-memory / LDS path counters of the tower convolutions, one hardware block per pass
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 300 python tools/_diag/stag_check.py > $out/r4b_stag_check.txt 2>&1 || { tail -20 $out/r4b_stag_check.txt; exit 1; }
cat $out/r4b_stag_check.txt
MFMA_ONE_TIME=1 MFMA_ONE_N=40 timeout -k 10 200 python tools/mfma_one.py > $out/r4b_standalone.txt 2>&1 || exit 1
cat $out/r4b_standalone.txt
for round in 1 2; do
  for stag in 0 1; do
    KA_CONV_P_STAG=$stag timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $out/r4b_bench_stag${stag}_$round.json 2> $out/r4b_bench_stag${stag}_$round.err || exit 1
    python - $out/r4b_bench_stag${stag}_$round.json $stag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("stag", sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("conv3x3_forward_launches_only"), d.get("wgrad_kernel"), flush=True)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
export MFMA_ONE=fwd,fwd2,dgrad,dgradm
pass() {
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/r4b_pmc_$tag -o c -- python3 $root/tools/mfma_one.py > $out/r4b_pmc_$tag.log 2>&1 || echo "pass $tag failed: $(grep -m1 'error code' $out/r4b_pmc_$tag.log)"
}
pass ta1 TA_BUSY_avr GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq1 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
cd $root
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in sorted(glob.glob(out + "/r4b_pmc_*")):
    if not d.endswith(("ta1", "ta2", "tcp1", "tcp2", "sq1", "sq2")): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "conv3x3" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k[:k.index("(")] if "(" in k else k
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
res = {k: {m: round(v / max(n[k][m], 1), 1) for m, v in c.items()} for k, c in acc.items()}
json.dump(res, open(out + "/r4b_vmem_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
