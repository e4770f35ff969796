#!/bin/bash
# secondary measurements of a round -> gpurun_out/<tag>_secondary.json   (tools/secondary.sh r01)
tag=${1:-r01}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --whole-update --sl-epoch --no-cpu-baseline > gpurun_out/${tag}_sec_bench.json 2> gpurun_out/${tag}_sec_bench.err
KA_SELECT_AMP_ONLY=0 timeout -k 10 300 python tools/select_bench.py > gpurun_out/${tag}_sec_select.txt 2>&1
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open(f"gpurun_out/{tag}_sec_bench.json").read().strip().splitlines()[-1])
out = {"command": "python bench.py --whole-update --sl-epoch --no-cpu-baseline ; python tools/select_bench.py",
       "headline_in_the_same_run": {k: d[k] for k in ("value", "unit", "ms_per_step")},
       "whole_update": d.get("whole_update"), "sl_epoch": d.get("sl_epoch"),
       "select_actions": [l.strip() for l in open(f"gpurun_out/{tag}_sec_select.txt") if l.startswith("amp=")]}
json.dump(out, open(f"gpurun_out/{tag}_secondary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
