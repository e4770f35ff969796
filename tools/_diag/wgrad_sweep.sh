for cfg in "X=1" "KA_WGRAD_WGS=128" "KA_WGRAD_WGS=160" "KA_WGRAD_WGS=224" "KA_WGRAD_WGS=256" "X=2"; do
  echo -n "$cfg: "
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" || exit 1
done
