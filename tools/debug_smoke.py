import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd.training.model_registry import build_model
from oracle import keisei_oracle as orc
torch.manual_seed(0)
C = int(os.environ.get("SM_C", 64)); Bn = int(os.environ.get("SM_B", 16))
shape = orc.NetShape(num_blocks=2, channels=C, se_reduction=8, global_pool_channels=32, policy_channels=16, value_fc_size=64, score_fc_size=32)
sd = orc.synth_state_dict(shape)
model = build_model("se_resnet", dict(shape.__dict__)); model.load_state_dict(sd); model.to("cuda:0").train()
mb = orc.synth_minibatch(Bn, seed=3)
out = model(mb["obs"].to("cuda:0"))
loss = (out.policy_logits ** 2).mean() + out.value_logits.sum() + out.score_lead.sum()
loss.backward()
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
live = dict(sd); live.update(leaves)
p2, v2, s2 = orc.seresnet_forward(live, mb["obs"], shape.num_blocks, train=True)
((p2 ** 2).mean() + v2.sum() + s2.sum()).backward()
print("fwd err", float((out.policy_logits.cpu() - p2.detach()).abs().max()) / float(p2.abs().max()))
for n, p in model.named_parameters():
    e = float((p.grad.cpu() - leaves[n].grad).abs().max()) / (float(leaves[n].grad.abs().max()) + 1e-9)
    l2 = float((p.grad.cpu() - leaves[n].grad).norm() / (leaves[n].grad.norm() + 1e-12))
    if e > 1e-3: print(f"{n:34s} max-rel {e:.3e}  rel-L2 {l2:.3e}")
