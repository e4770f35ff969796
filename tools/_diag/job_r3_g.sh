#!/bin/bash
# round 3, job G: transformer step profile (kernel trace, per-shape GEMM durations) + the new GPU env test
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_shogi_env.py -x -q -m gpu -k "python_api or reference_vector" > gpurun_out/r3_g_tests.txt 2>&1; tail -2 gpurun_out/r3_g_tests.txt
timeout -k 10 300 python bench.py --workload transformer --no-cpu-baseline > gpurun_out/r3_tf_line.json 2>gpurun_out/r3_tf.err; cut -c1-300 gpurun_out/r3_tf_line.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_tf_trace -o t -- python3 $R/bench.py --workload transformer --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3_tf_prof.json 2> $R/gpurun_out/r3_tf_prof.err
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_tf_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.index("(")] if "(" in name else name
    key = (name[:40], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
tot = sum(v[1] for v in agg.values())
out = open("gpurun_out/r3_tf_by_shape.txt", "w")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    line = f"{k[0]:40s} grid {k[1]:>8s} {k[2]:>5s} {k[3]:>3s}  calls {v[0]:4d}  total {v[1]/1e3:8.2f} ms  avg {v[1]/v[0]:8.1f} us  {100*v[1]/tot:5.1f} %"
    print(line); out.write(line + "\n")
PY
