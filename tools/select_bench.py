"""Latency of KataGoPPOAlgorithm.select_actions (rollout inference path) at N environments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams
from keisei_amd.training.model_registry import build_model
dev = torch.device("cuda")
for amp in ((True,) if os.environ.get("KA_SELECT_AMP_ONLY", "1") == "1" else (True, False)):
    model = build_model("se_resnet", dict(num_blocks=40, channels=256, se_reduction=16, global_pool_channels=128,
                                          policy_channels=32, value_fc_size=256, score_fc_size=128, obs_channels=50)).to(dev)
    algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=4096, use_amp=amp), model)
    for N in (128, 512, 2048):
        obs = torch.randn(N, 50, 9, 9, device=dev)
        masks = torch.zeros(N, 11259, dtype=torch.bool, device=dev); masks[:, :3753] = True
        for _ in range(3): algo.select_actions(obs, masks)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): out = algo.select_actions(obs, masks)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"amp={amp} N={N}: {dt * 1e3:.2f} ms per call, {N / dt:.0f} positions/s", flush=True)
