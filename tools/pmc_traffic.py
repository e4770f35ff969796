"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/*.json.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<workload note>"
Correction per MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read stream, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, json, re, sys

def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        agg[name].append(float(r["Counter_Value"]))
    return agg

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two separate passes) --kernel-trace -- python3 bench.py "
                  "--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events",
       "correction": "counters are in KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of a wide coalesced read stream); WRITE_SIZE exact",
       "workload": sys.argv[4] if len(sys.argv) > 4 else "", "kernels": {}}
# the build the counters were collected on (bench.py refuses to report traffic from a profile of other kernel sources)
sys.path.insert(0, ".")
import bench  # noqa: E402
out["kernel_source_sha16"] = bench.kernel_source_id()
out["transformer_source_sha16"] = bench.kernel_source_id(("common.h", "transformer.hip"))
for k in sorted(set(fetch) | set(write)):
    f = 2.0 * 1024.0 * sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
    w = 1024.0 * sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
    out["kernels"][k] = {"launches": len(fetch.get(k, write.get(k, []))), "fetch_bytes_per_launch_corrected": int(f),
                         "write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
