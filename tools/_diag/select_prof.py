import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams
from keisei_amd.training.model_registry import build_model
dev = torch.device("cuda")
model = build_model("se_resnet", dict(num_blocks=40, channels=256, se_reduction=16, global_pool_channels=128,
                                      policy_channels=32, value_fc_size=256, score_fc_size=128, obs_channels=50)).to(dev)
algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=4096, use_amp=True), model)
N = 128
obs = torch.randn(N, 50, 9, 9, device=dev); masks = torch.zeros(N, 11259, dtype=torch.bool, device=dev); masks[:, :3753] = True
for _ in range(6): algo.select_actions(obs, masks)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): algo.select_actions(obs, masks)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
