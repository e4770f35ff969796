#!/bin/bash
# rollout-inference latency under a few runtime switches, one job: tools/select_ab.sh
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/select_bench.py 2>&1 | grep "amp=True"; }
run KA_EVAL_GRAPH_FORK=0
run KA_EVAL_GRAPH_FORK=1
run KA_EVAL_GRAPH_FORK=0 KA_CONV_WM=2
run KA_EVAL_GRAPH_FORK=1 KA_CONV_WM=2
run KA_EVAL_GRAPH_FORK=1 KA_CONV_NTW=1
run KA_EVAL_GRAPH_FORK=1 KA_CONV_NTW=4
