"""Stand-alone timing of the one-launch eval tower (csrc/tower.hip) at rollout batch sizes; KA_TOWER_ABL bit flags
(diagnostics: wrong results) switch phases off to see what each costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
from keisei_amd.training.model_registry import build_model
dev = torch.device("cuda")
model = build_model("se_resnet", dict(num_blocks=40, channels=256, se_reduction=16, global_pool_channels=128,
                                      policy_channels=32, value_fc_size=256, score_fc_size=128, obs_channels=50)).to(dev).eval()
model.configure_amp(True, torch.bfloat16, "cuda")
os.environ["KA_EVAL_GRAPH"] = "0"
with torch.no_grad():
    model(torch.randn(4, 50, 9, 9, device=dev))
eng = model._hip_engine
tab = list(eng._tower_tabs.values())[-1]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for N in [int(v) for v in (sys.argv[1:] or ['128', '256', '512'])]:
    x = torch.randn(N, 81, 256, device=dev).bfloat16(); pool = torch.randn(N, 1024, device=dev)
    xo = torch.empty_like(x); po = torch.empty_like(pool)
    for abl in [0] + ([1, 2, 3, 4, 8, 12, 15] if N in (128, 4096) else []):
        os.environ["KA_TOWER_ABL"] = str(abl)
        ms = timeit(lambda: _lib.call("ka_tower_eval", x, pool, xo, po, tab, 40, N, 256, 128, 16, 1, _lib.stream_ptr()))
        print(f"N={N} abl={abl:2d}: {ms * 1e3:8.1f} us per launch, {ms * 1e3 / 40:6.1f} us per block", flush=True)
os.environ.pop("KA_TOWER_ABL", None)
