"""GPU: bench.py prints ONE JSON line with the driver's contract keys (run on the small 6x128 workload to stay short)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(*args):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench("--workload", "6x128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["unit"] == "samples/s" and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches_timed"] > 0
    assert d["cpu_baseline"] is None                    # skipped on request; the default run fills it (rank 0, N = 1)
    assert all(v == v for v in d["train_metrics"].values())   # finite training metrics: the timed steps really trained
