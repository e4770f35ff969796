"""Per-kernel roofline table of one bench run: average launch time (rocprofv3 --stats) x HBM bytes per launch (--pmc passes)
-> achieved GB/s, and TFLOP/s for the MFMA kernels.   usage: roofline_table.py <kernel_stats.csv> <pmc.json> <out.json>"""
import csv, json, re, sys

stats = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "")
    stats[re.sub(r"\(.*", "", name)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
pmc = json.load(open(sys.argv[2]))["kernels"]
FLOP = {"conv3x3_kernel<bf16_t, 4, 1>": 391.4e9, "wgrad_kernel<bf16_t, 128, false>": 391.4e9, "wgrad_kernel<bf16_t, 128, true>": 391.4e9}
rows = []
for k, (calls, us, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.4:
        continue
    rec = pmc.get(k)
    row = {"kernel": k, "launches_in_run": calls, "avg_us": round(us, 1), "percent_of_kernel_time": pct}
    if rec:
        row["hbm_MB_per_launch"] = round(rec["hbm_bytes_per_launch"] / 1e6, 1)
        row["achieved_GBps"] = round(rec["hbm_bytes_per_launch"] / us / 1e3, 0)
        row["frac_of_8TBps"] = round(rec["hbm_bytes_per_launch"] / us / 1e3 / 8000, 3)
    if k in FLOP:
        row["achieved_TFLOPs"] = round(FLOP[k] / us / 1e6, 0)
        row["frac_of_2500"] = round(FLOP[k] / us / 1e6 / 2500, 3)
    rows.append(row)
json.dump({"note": "in-step averages (kernels share the chip with the weight-gradient stream); HBM bytes = FETCH_SIZE x 2 + "
                   "WRITE_SIZE per launch from separate --pmc passes of the same workload", "kernels": rows},
          open(sys.argv[3], "w"), indent=1)
for r in rows:
    print(r)
