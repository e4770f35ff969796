"""Main-queue composition and gaps of the last forward / backward of a rocprofv3 kernel trace.
usage: timeline.py <kernel_trace.csv>"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:46]
find = lambda pat: [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
a, b, c = find("obs_to_nhwc")[-1], find("policy_loss_kernel")[-1], find("adam_kernel")[-1]
for title, seg in (("forward", rows[a:b + 1]), ("backward", rows[b:c + 1])):
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    print(f"== {title}: span {(t1 - t0) / 1e6:.2f} ms, {len(seg)} kernels")
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
    for qid, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        agg, gap, prev = collections.defaultdict(lambda: [0, 0]), 0, None
        for r in rs:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += e - s
            if prev is not None:
                gap += max(0, s - prev)
            prev = max(prev or 0, e)
        busy = sum(v[1] for v in agg.values())
        print(f"  queue {qid}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, gaps {gap / 1e6:.2f} ms")
        for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"    {k:48s} {n:4d} {d / 1e6:7.3f} ms  avg {d / n / 1e3:7.1f} us")
