"""One self-play rollout step at N games (SURVEY 8 f1 + f2 + f3 together): select_actions on the 40x256 network, VecEnv.step,
buffer.add -- with the device env handing out tensors ("device"), and with the env's results taken the way the reference's loop
takes them from its CPU env ("host hand-over": numpy results, torch.from_numpy(...).to(device), actions.tolist()); the env
itself runs on the GPU in both, so the difference is the PCIe / host hand-over alone.  Writes gpurun_out/<tag>_rollout_step.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from keisei_amd.shogi_gym import VecEnv
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer
from keisei_amd.training.model_registry import build_model

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
dev = torch.device("cuda")
model = build_model("se_resnet", dict(num_blocks=40, channels=256, se_reduction=16, global_pool_channels=128,
                                      policy_channels=32, value_fc_size=256, score_fc_size=128, obs_channels=50)).to(dev)
algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=4096, use_amp=True), model)
out = {"model": "se_resnet 40x256, bf16", "cases": []}
for N in (128, 512):
    for mode in ("device", "host hand-over"):
        torch_out = mode == "device"
        env = VecEnv(num_envs=N, max_ply=500, observation_mode="katago", action_mode="spatial",
                     output="torch" if torch_out else "numpy", check_actions=not torch_out)
        buf = KataGoRolloutBuffer(N, (50, 9, 9), 11259)
        r = env.reset()
        to_dev = (lambda x: x) if torch_out else (lambda x: torch.from_numpy(x).to(dev))
        obs, legal = to_dev(r.observations), to_dev(r.legal_masks)
        steps, warm = 40, 8
        for s in range(steps + warm):
            if s == warm:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            actions, logp, values = algo.select_actions(obs, legal)
            res = env.step(actions if torch_out else actions.tolist())
            rew, term, trunc = to_dev(res.rewards), to_dev(res.terminated), to_dev(res.truncated)
            done = term | trunc
            cats = torch.where(done, torch.where(rew > 0, 0, torch.where(rew < 0, 2, 1)), -1)
            score = to_dev(res.step_metadata.material_balance).float() / 76.0
            buf.add(obs, actions, logp, values, rew, done.float(), term.float(), legal, cats, score)
            obs, legal = to_dev(res.observations), to_dev(res.legal_masks)
            if buf.size >= 32 * N:
                buf.clear()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out["cases"].append({"games": N, "env_results": mode, "ms_per_rollout_step": dt * 1e3, "positions_per_s": N / dt})
        print(out["cases"][-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{tag}_rollout_step.json", "w"), indent=1)
