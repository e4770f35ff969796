"""Diff of every saved forward tensor and every gradient between two KA_CONV_KC settings (fp32 mid model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import Golden
from keisei_amd.training.models.se_resnet import SEResNetModel, SEResNetParams
from keisei_amd.hip.seresnet import SEResNetEngine
from keisei_amd import _lib
from oracle import keisei_oracle as orc
g = Golden("g2_model_mid")
tag, shape = "s3x256.", orc.NetShape(3, 256)
m = SEResNetModel(SEResNetParams(**shape.__dict__))
m.load_state_dict(orc.synth_state_dict(shape), strict=True)
m.cuda().train()
for mod in m.modules():
    if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm): mod.momentum = 0.0
obs = g[tag + "obs"].cuda()
eng = SEResNetEngine(m)
B = obs.shape[0]
def run(kc):
    if kc: os.environ["KA_CONV_KC"] = str(kc); _lib.reload_options()
    else: os.environ.pop("KA_CONV_KC", None)
    logits, v, s, sv = eng.forward(obs, True, True, torch.float32)
    grads = eng.backward(sv, g[tag + "cot.policy"].cuda() / B, g[tag + "cot.value"].cuda(), g[tag + "cot.score"].cuda())
    torch.cuda.synchronize()
    flat = {"logits": logits, "v": v, "s": s}
    names = "bx bpool y1 sc1 sh1 mu1 is1 g1 g y2 sc2 sh2 mu2 is2 sqz se1 se out".split()
    for i, blk in enumerate(sv.blocks):
        for n, t in zip(names, blk): flat[f"b{i}.{n}"] = t
    for n, t in grads.items(): flat["grad." + n] = t
    return {k: t.detach().float().cpu().clone() for k, t in flat.items()}
a, b = run(128), run(64)
for k in a:
    d = float((a[k] - b[k]).abs().max()); s = float(a[k].abs().max())
    if d > 1e-5 * s: print(f"{k:40s} maxdiff {d:.3e} scale {s:.3e} rel {d / (s + 1e-30):.2e}")
print("done")
for i in range(3):
    fo = int(((a[f"b{i}.out"] > 0) != (b[f"b{i}.out"] > 0)).sum())
    ha = a[f"b{i}.y1"] * a[f"b{i}.sc1"] + a[f"b{i}.sh1"]; hb = b[f"b{i}.y1"] * b[f"b{i}.sc1"] + b[f"b{i}.sh1"]
    f1 = int(((ha > 0) != (hb > 0)).sum())
    print(f"block {i}: out-mask flips {fo}, bn1-relu flips {f1}, of {a[f'b{i}.out'].numel()}")
d = (a["grad.blocks.2.se_fc2.bias"] - b["grad.blocks.2.se_fc2.bias"]).abs()
print("se_fc2.bias elements differing >1e-4*scale:", int((d > 1e-4 * a["grad.blocks.2.se_fc2.bias"].abs().max()).sum()), "of", d.numel())
