#!/bin/bash
# three-way A/B inside one job: this build, then the libraries given as arguments, twice round-robin
for round in 1 2; do
  for l in keisei_amd/libkeisei_amd.so "$@"; do
    echo -n "$l round $round: "
    KEISEI_AMD_LIB=$l timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
