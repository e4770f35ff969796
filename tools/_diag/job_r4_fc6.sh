#!/bin/bash
# fc_chain_kernel phase 1: ablation (wrong results) -- the lanes of a weight load read contiguous memory -- per-wave stamps, cold L2
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
for v in fctl fccon; do
  echo "== $v"; KEISEI_AMD_LIB=$PWD/keisei_amd/libka_$v.so timeout -k 10 200 python tools/_diag/fc_chain_tl.py 2>&1 | grep -v amdgpu | grep -A1 "gpool fwd" | cut -c1-250
done > $out/r4_fc_ablation2.txt 2>&1
cat $out/r4_fc_ablation2.txt
