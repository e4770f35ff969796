import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import Golden
import test_hip_ppo as T
g = Golden("g5_update")
class MP:
    def setattr(self, obj, name, val): setattr(obj, name, val)
gg, m, algo, buf = T.make(lambda n: g)
it = iter(list(g["perms"]))
orig = torch.randperm
torch.randperm = lambda n, *a, device=None, **k: next(it).to(device or "cpu")
from keisei_amd.training.value_adapter import MultiHeadValueAdapter
met = algo.update(buf, g["next_values"].cuda(), value_adapter=MultiHeadValueAdapter(1.5, 0.1, 0.1))
torch.randperm = orig
ref = g.sub("sd1."); ref0 = g.sub("sd0."); got = m.state_dict()
for k, v in ref.items():
    if v.dtype.is_floating_point:
        d = (got[k].cpu() - v).abs()
        upd = (v - ref0[k]).abs().max()
        print(f"{k:32s} maxdiff {float(d.max()):.2e}  n>3e-5: {int((d > 3e-5).sum())}/{d.numel()}  max update {float(upd):.2e}")
