"""Where does one update() spend the time that is not minibatch steps?  (bench.py's `secondary.whole_update` leg, device store,
each phase bracketed by a device synchronisation; two updates, the second without first-use costs.)"""
import os, sys, time, types, argparse
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
from keisei_amd.training.katago_ppo import KataGoRolloutBuffer
args = argparse.Namespace(workload="40x256", batch=0)
device = torch.device("cuda:0")
torch.cuda.set_device(0)
res = bench.run_workload(args, "bf16", 2, 2, device, 0, 1, 0, 0)
algo, adapter = res["algo"], res["adapter"]
nb, C, Rr, G, P, V, S, T, N = res["shape"]
A = 11259
acc = {}
def timed(obj, name):
    orig = getattr(obj, name)
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, wrap)
for n in ("_advantages", "_fused_begin", "_fused_step", "_fused_end"):
    timed(algo, n)
os.environ["KA_ROLLOUT_BUFFER"] = "device"
for rep in range(2):
    g = torch.Generator().manual_seed(99)
    buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
    timed(buf, "flatten_packed"); timed(buf, "clear")
    masks = torch.zeros(N, A, dtype=torch.bool); masks[:, : A // 3] = True; masks = masks.to(device)
    for t in range(T):
        last = t == T - 1
        done = torch.full((N,), last, dtype=torch.bool)
        cats = torch.randint(0, 3, (N,), generator=g) if last else torch.full((N,), -1, dtype=torch.long)
        step = [torch.randn(N, 50, 9, 9, generator=g), torch.randint(0, A // 3, (N,), generator=g), -8.2 + 0.05 * torch.randn(N, generator=g),
                torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g), done, done, None, cats, torch.randn(N, generator=g).clamp(-1.5, 1.5)]
        buf.add(*[masks if v is None else v.to(device) for v in step])
    nv = torch.randn(N, generator=g).to(device)
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    algo.update(buf, nv, value_adapter=adapter)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    named = sum(acc.values())
    print(f"update {rep}: {dt * 1e3:.1f} ms total for {T * N} transitions x {algo.params.epochs_per_batch} epochs;  " +
          "  ".join(f"{k} {v * 1e3:.1f}" for k, v in acc.items()) + f"  unaccounted {1e3 * (dt - named):.1f}", flush=True)
