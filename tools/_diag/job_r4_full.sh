#!/bin/bash
# full GPU suite + smoke + default bench line on the final sources
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/full_gpu_tests.txt; exit 1; }
tail -3 gpurun_out/full_gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.txt 2>&1 || { tail -20 gpurun_out/full_smoke.txt; exit 1; }
tail -3 gpurun_out/full_smoke.txt
timeout -k 10 600 python bench.py > gpurun_out/full_bench_default.json 2> gpurun_out/full_bench_default.err || { tail -20 gpurun_out/full_bench_default.err; exit 1; }
tail -1 gpurun_out/full_bench_default.json | cut -c1-400
