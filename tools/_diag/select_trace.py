"""Composition of the last select_actions call in a rocprofv3 kernel trace of tools/select_one.py (argv[1])."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'obs_to_nhwc' in r['Kernel_Name']]
seg = rows[starts[-1]:]
t0, t1 = seg[0]['s'], max(r['e'] for r in seg)
print(f"last call: span {(t1 - t0) / 1e3:.1f} us, {len(seg)} kernels")
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:56]
    agg[k][0] += r['e'] - r['s']; agg[k][1] += 1
busy = sum(v[0] for v in agg.values())
print(f"busy (all queues) {busy / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {k:58s} {v[1]:4d} {v[0] / 1e3:8.1f} us  avg {v[0] / v[1] / 1e3:6.1f}")
