#!/bin/bash
# round 4, final evidence on the shipped kernels: default bench line, kernel stats, PMC traffic, timeline, roofline table,
# transformer stats + traffic (tools/profile_round.sh), SQ / GRBM counters of the MFMA forms (tools/mfma_counters.sh)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
timeout -k 10 900 python bench.py > $out/r04_bench_default_line.json 2> $out/r04_bench_default_line.err || { tail -5 $out/r04_bench_default_line.err; exit 1; }
tail -1 $out/r04_bench_default_line.json | cut -c1-300
bash tools/profile_round.sh r04 > $out/r04_profile_round.log 2>&1 || { tail -20 $out/r04_profile_round.log; exit 1; }
tail -4 $out/r04_profile_round.log | cut -c1-400
bash tools/mfma_counters.sh r04 > $out/r04_mfma_counters.log 2>&1 || { tail -5 $out/r04_mfma_counters.log; exit 1; }
tail -2 $out/r04_mfma_counters.log | cut -c1-300
