"""Stand-alone throughput of the device VecEnv (shogi_env.hip): env-steps/s at a few game counts with on-device action
sampling excluded (actions are drawn beforehand from the masks of a recorded playout is impossible -- masks depend on the
moves -- so the sampler runs between steps and only the env launches are timed with HIP events), next to the CPU oracle on
one host core.  Writes gpurun_out/<tag>_env_steps.json."""
import json, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from keisei_amd.shogi_gym import VecEnv

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
out = {"bytes_written_per_env_step": 50 * 81 * 4 + 11259 + 352 * 4 + 128, "cases": []}
for n in (128, 512, 4096, 16384):
    env = VecEnv(num_envs=n, max_ply=500, observation_mode="katago", action_mode="spatial", output="torch", check_actions=False)
    r = env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    steps, tot = 200, 0.0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for s in range(steps + 20):
        acts = torch.multinomial(r.legal_masks.float(), 1, generator=g).squeeze(1)
        a.record(); r = env.step(acts); b.record(); torch.cuda.synchronize()
        if s >= 20: tot += a.elapsed_time(b)
    env.raise_if_refused()
    ms = tot / steps
    out["cases"].append({"envs": n, "ms_per_step": ms, "env_steps_per_s": n / ms * 1e3,
                         "written_GBps": n * out["bytes_written_per_env_step"] / ms / 1e6,
                         "mean_legal_moves": float(r.legal_masks.sum(1).float().mean())})
    print(out["cases"][-1], flush=True)
os.environ["OMP_NUM_THREADS"] = "1"          # the recorded baseline is one host core (the oracle steps games in parallel under OpenMP otherwise)
from oracle.shogi import OracleVecEnv
e = OracleVecEnv(128, 500); obs, mask = e.reset(); rng = np.random.default_rng(0)
acts_t = 0.0; t0 = time.time(); k = 0
while time.time() - t0 < 12:
    t1 = time.time(); acts = [int(rng.choice(np.flatnonzero(m))) for m in mask]; acts_t += time.time() - t1
    mask = e.step(acts)["legal_masks"]; k += 1
dt = time.time() - t0 - acts_t
out["cpu_oracle"] = {"env_steps_per_s": 128 * k / dt, "cores": 1, "kind": "port", "sample": f"{k} steps of 128 games, random legal play"}
print(out["cpu_oracle"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{tag}_env_steps.json", "w"), indent=1)
