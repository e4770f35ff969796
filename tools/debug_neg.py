import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_fullsize import conv, pack, B, C, DEV
dt = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(5)
x1 = torch.randn(B, 81, C, device=DEV, generator=g).to(dt)
w = torch.randn(C, C, 3, 3, device=DEV, generator=g) / 48.0
wp = pack(w, dt, 0)
y1, _, _ = conv(x1, wp, dt)
y1b, _, _ = conv(x1, wp, dt)
print("repeat equal:", torch.equal(y1, y1b))
yn, _, _ = conv((-x1.float()).to(dt), wp, dt, stats=False)
d = (yn.float() + y1.float()).abs()
print("neg: frac differing", float((d > 0).float().mean()), "max abs", float(d.max()), "max rel", float((d / (y1.float().abs() + 1e-6)).max()))
idx = (d > 0).nonzero()[:5]
for i in idx: print(i.tolist(), float(y1[tuple(i)]), float(yn[tuple(i)]))
