"""Per-stream busy time / gaps of the last bench step in a rocprofv3 kernel trace CSV (argv[1])."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0, t1 = step[0]['s'], step[-1]['e']
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(step))
by = collections.defaultdict(list)
for r in step: by[r['Stream_Id']].append(r)
for k, v in by.items():
    print("stream", k, "n", len(v), "busy ms", sum(r['e'] - r['s'] for r in v) / 1e6)
main = max(by.values(), key=len)
pl = [r for r in step if 'policy_loss' in r['Kernel_Name']][0]
print("forward ms", (pl['s'] - t0) / 1e6, "backward+opt ms", (t1 - pl['s']) / 1e6)
gaps = collections.defaultdict(lambda: [0, 0])
for a, b in zip(main[:-1], main[1:]):
    g = b['s'] - a['e']
    if g > 0:
        k = b['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        gaps[k][0] += g; gaps[k][1] += 1
print("main gaps total ms", sum(v[0] for v in gaps.values()) / 1e6)
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:10]: print(f"  gap before {k:40s} {v[0] / 1e6:7.2f} ms over {v[1]}")
for sid, v in by.items():
    agg = collections.defaultdict(lambda: [0, 0])
    for r in v:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]
        agg[k][0] += r['e'] - r['s']; agg[k][1] += 1
    print("stream", sid)
    for k, x in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]: print(f"  {k:50s} {x[0] / 1e6:7.2f} ms n={x[1]:4d} avg {x[0] / x[1] / 1e3:7.1f} us")
