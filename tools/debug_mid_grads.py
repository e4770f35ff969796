"""Gradient-norm ratios of the mid fp32 models against the golden fixture (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import Golden
from keisei_amd.training.models.se_resnet import SEResNetModel, SEResNetParams
from oracle import keisei_oracle as orc
g = Golden("g2_model_mid")
tag, shape = "s3x256.", orc.NetShape(3, 256)
for rep in range(int(os.environ.get("REPS", 3))):
    m = SEResNetModel(SEResNetParams(**shape.__dict__))
    m.load_state_dict(orc.synth_state_dict(shape), strict=True)
    m.cuda()
    obs = g[tag + "obs"].cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm): mod.momentum = 0.0
    o = m(obs)
    B = obs.shape[0]
    loss = ((o.policy_logits * g[tag + "cot.policy"].cuda()).sum() / B + (o.value_logits * g[tag + "cot.value"].cuda()).sum()
            + (o.score_lead * g[tag + "cot.score"].cuda()).sum())
    loss.backward()
    names = list(g.np(tag + "grad_names")); norms = dict(zip(names, g.np(tag + "grad_norms")))
    grads = dict((n, p.grad) for n, p in m.named_parameters())
    bad = [(n, float(grads[n].double().norm()) / norms[n]) for n in names if abs(float(grads[n].double().norm()) / norms[n] - 1) > 2e-3]
    print(f"rep {rep}: B={B} bad={bad}", flush=True)
