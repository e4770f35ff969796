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


def test_bench_gpus_2_starts_itself():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's recorded `bench.py --gpus 1 ...`):
    the process spawns its two ranks itself (the reference's run.sh:309-310 does it with torchrun), here both on the one
    GPU of the box over gloo (KA_BENCH_BACKEND=gloo; RCCL needs one device per rank); ONE line, n_gpus 2, value = sum of ranks."""
    env_backend = os.environ.get("KA_BENCH_BACKEND")
    os.environ["KA_BENCH_BACKEND"] = "gloo"
    try:
        d = run_bench("--gpus", "2", "--workload", "2x32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    finally:
        if env_backend is None:
            os.environ.pop("KA_BENCH_BACKEND", None)
        else:
            os.environ["KA_BENCH_BACKEND"] = env_backend
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["config"]["parallelism"] == "dp2+syncbn" and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]   # both ranks' samples
    assert all(v == v for v in d["train_metrics"].values())


def test_default_line_carries_the_secondary_measurements():
    """The driver runs `python bench.py` with no workload flag: SURVEY 8d's "report both" (whole update() beside the minibatch step)
    and BASELINE configs[1] / configs[4] ride on that line as bounded legs (`secondary`), next to the headline's own keys."""
    d = run_bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-fp32")
    assert d["metric"].startswith("PPO samples/sec, se_resnet 40x256") and d["roofline"]["launches_timed"] > 0
    sec = d["secondary"]
    assert set(sec) == {"whole_update", "workload_6x128", "transformer"}
    assert sec["whole_update"]["samples_per_s"] > 0 and sec["whole_update"]["store"] == "device"
    assert sec["whole_update"]["transitions"] == 128 * 128
    for k in ("workload_6x128", "transformer"):
        assert sec[k]["samples_per_s"] > 0 and 3 <= sec[k]["steps"] <= 20 and "workload" in sec[k]
        r = sec[k]["roofline"]
        assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_gpus_4_rehearsal_over_gloo():
    """The launcher at the largest rank count this pool lets one card carry beside the test process itself (six processes may
    have a GPU open at once; the eight-rank wiring of BASELINE configs[3] is rehearsed on CPU by
    tests/test_ddp_gloo.py::test_eight_rank_ddp_update_keeps_ranks_in_sync): `bench.py --gpus 4` over gloo on the one GPU, every
    rank seen, dp4 + SyncBatchNorm, the value the sum of the ranks."""
    saved = {k: os.environ.get(k) for k in ("KA_BENCH_BACKEND",)}
    os.environ["KA_BENCH_BACKEND"] = "gloo"
    try:
        d = run_bench("--gpus", "4", "--workload", "2x32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert d["n_gpus"] == 4 and d["n_ranks_seen"] == 4 and d["backend"] == "gloo"
    assert d["config"]["parallelism"] == "dp4+syncbn" and d["config"]["global_batch"] == 4 * d["config"]["per_gpu_batch"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]
    assert all(v == v for v in d["train_metrics"].values())
