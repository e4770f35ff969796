"""Random-shape sweep of the conv3x3 / wgrad / FC kernels against fp32 torch on the CPU (run on an MI355X)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from keisei_amd import _lib
DEV = "cuda"
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
st = lambda: _lib.stream_ptr()
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    dtn = random.choice(["bf16", "f32"])
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    cpk = 32 if dtn == "bf16" else 16
    B = random.choice([1, 2, 3, 5, 8, 17, 64, 130, 257, 600])
    cin = cpk * random.randint(1, 10)
    cout = 16 * random.randint(1, 18)
    if dtn == "bf16":
        cout = 32 * random.randint(1, 9)          # dgrad of a bf16 layer needs Cout % 32
    code = _lib.dtype_code(dt)
    g = torch.Generator().manual_seed(trial)
    x = torch.randn(B, cin, 9, 9, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    rnd = lambda t: t.to(dt).float()
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], 81, t.shape[1]).contiguous().to(dt).to(DEV)
    back = lambda t, c: t.float().cpu().reshape(-1, 9, 9, c).permute(0, 3, 1, 2)
    ref = F.conv2d(rnd(x), rnd(w), padding=1)
    wp = torch.empty(9 * (cin // cpk) * (cout // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w.to(DEV), wp, cout, cin, cout, cin, 0, code, st())
    out = torch.empty(B, 81, cout, dtype=dt, device=DEV)
    bsum = torch.empty(B, cout, device=DEV); sq = torch.empty(_lib.query("ka_conv3x3_sqpart_rows", B), cout, device=DEV)
    _lib.call("ka_conv3x3_fwd", nhwc(x), wp, out, None, None, None, 0, bsum, sq, B, cin, cout, code, st())
    tol = (2e-2 if dtn == "bf16" else 1e-4) * float(ref.abs().max())
    e1 = float((back(out, cout) - ref).abs().max())
    e2 = float((bsum.cpu() - ref.sum(dim=(2, 3))).abs().max())
    dy = torch.randn(B, cout, 9, 9, generator=g) / 8
    refw = torch.nn.grad.conv2d_weight(rnd(x), (cout, cin, 3, 3), rnd(dy), padding=1)
    ns = _lib.query("ka_wgrad_splits", B, cin, cout, 0)
    slab = torch.empty(ns * 9 * cout * cin, device=DEV); dw = torch.empty(cout, cin, 3, 3, device=DEV)
    _lib.call("ka_conv3x3_wgrad", nhwc(dy), nhwc(x), None, None, None, 0, slab, dw, B, cin, cin, cout, 0, 0, code, st())
    e3 = float((dw.cpu() - refw).abs().max()) / (float(refw.abs().max()) + 1e-9)
    ok = e1 <= tol and e2 <= 81 * tol and e3 <= (3e-2 if dtn == "bf16" else 1e-4)
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {dtn} B={B} Cin={cin} Cout={cout}: conv {e1:.2e} (tol {tol:.2e}) bsum {e2:.2e} wgrad rel {e3:.2e}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
