import sys; sys.path.insert(0, '.')
import numpy as np, torch
from keisei_amd.training.models.se_resnet import SEResNetModel, SEResNetParams
from oracle import keisei_oracle as orc
z = np.load('tests/golden/g2_model_mid.npz')
tag, shape = "s6x128.", orc.NetShape(6, 128)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sd = orc.synth_state_dict(shape)
if len(sys.argv) > 2 and sys.argv[2] == 'default':
    torch.manual_seed(1); sd = {k: v.clone() for k, v in SEResNetModel(SEResNetParams(**shape.__dict__)).state_dict().items()}
g = torch.Generator().manual_seed(5)
obs = torch.randn(B, 50, 9, 9, generator=g)
cp, cv, cs = torch.randn(B, 9, 9, 139, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, 1, generator=g)
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
live = dict(sd); live.update(leaves)
p, v, s = orc.seresnet_forward(live, obs, shape.num_blocks, train=True, momentum=0.0)
((p * cp).sum() / B + (v * cv).sum() + (s * cs).sum()).backward()
for mode in ("bf16", "f32"):
    m = SEResNetModel(SEResNetParams(**shape.__dict__)); m.load_state_dict(sd); m.cuda().train()
    if mode == "bf16": m.configure_amp(True, torch.bfloat16, "cuda")
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d): mod.momentum = 0.0
    o = m(obs.cuda())
    ((o.policy_logits * cp.cuda()).sum() / B + (o.value_logits * cv.cuda()).sum() + (o.score_lead * cs.cuda()).sum()).backward()
    rows = []
    for n, prm in m.named_parameters():
        ref, got = leaves[n].grad.flatten().double(), prm.grad.flatten().double().cpu()
        if float(ref.norm()) == 0: continue
        rows.append((float((ref * got).sum() / (ref.norm() * got.norm() + 1e-30)), float(got.norm() / ref.norm()), n, ref.numel()))
    rows.sort()
    print(mode, "B", B)
    for r in rows[:10]: print("   cos %.4f ratio %.3f  %s (%d)" % r)
