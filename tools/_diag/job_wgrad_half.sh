mkdir -p gpurun_out
V=${1:-4}
WG_AB_V=$V timeout -k 10 300 python tools/wgrad_ab.py > gpurun_out/r3_wgrad_ab$V.txt 2>&1; cat gpurun_out/r3_wgrad_ab$V.txt | tail -13 | cut -c1-220
for v in 1 $V 1 $V; do KA_WGRAD_V=$v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 > gpurun_out/r3_ab_v$v.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r3_ab_v$v.json').read().strip().splitlines()[-1]);print('V=$v',d['value'],d['ms_per_step'],d['wgrad_kernel'],d['roofline']['avg_launch_ms'])"; done
cd /tmp && export TMPDIR=/tmp
KA_WGRAD_V=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_v${V}_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/r3_v${V}_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3_v${V}_prof.err
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find gpurun_out/r3_v${V}_stats -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_v${V}_timeline.txt; grep -A12 "== backward" gpurun_out/r3_v${V}_timeline.txt; grep "queue 2" -A4 gpurun_out/r3_v${V}_timeline.txt | tail -5
