"""Stand-alone time of the chained FC kernel (ka_fc_chain / ka_fc_chain_bwd) at the shapes of the 40x256 step, 4096 boards:
global-pool bias (768 -> 128 -> 256), squeeze-excite (256 -> 16 -> 512, BatchNorm affine on the input), and their backward
chains.  Isolated launches (a device-wide sync between launches: the kernel sits on a dependency chain in the step, so its
latency is what counts) and back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
dev, M = "cuda", int(os.environ.get("CB_B", 4096))
st = _lib.stream_ptr
R = lambda *s: torch.randn(*s, device=dev)
def fwd(K1, H, N2, affine):
    x, W1, b1, W2, b2 = R(M, K1), R(H, K1) / K1 ** 0.5, R(H), R(N2, H) / H ** 0.5, R(N2)
    sc, sh = (torch.rand(K1, device=dev) + 0.5, R(K1)) if affine else (None, None)
    xo = torch.empty(M, K1, device=dev) if affine else None
    hid, y = torch.empty(M, H, device=dev), torch.empty(M, N2, device=dev)
    return lambda: _lib.call("ka_fc_chain", x, sc, sh, 1.0 / 81, W1, b1, W2, b2, xo, hid, y, M, K1, K1, H, N2, st())
def bwd(N2, H, K1):
    dy, hid, W2T, W1T = R(M, N2), R(M, H), R(H, N2), R(K1, H)
    dh, dx = torch.empty(M, H, device=dev), torch.empty(M, K1, device=dev)
    return lambda: _lib.call("ka_fc_chain_bwd", dy, hid, W2T, W1T, dh, dx, M, N2, H, K1, st())
cases = {"gpool fwd 768-128-256": fwd(768, 128, 256, False), "se fwd 256-16-512 (affine)": fwd(256, 16, 512, True),
         "gpool bwd 256-128-768": bwd(256, 128, 768), "se bwd 512-16-256": bwd(512, 16, 256)}
for name, fn in cases.items():
    for _ in range(5): fn()
    torch.cuda.synchronize()
    iso = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        iso.append(a.elapsed_time(b) * 1e3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fn()
    b.record(); torch.cuda.synchronize()
    iso.sort()
    print(f"{name:28s} isolated median {iso[len(iso) // 2]:6.1f} us   back-to-back {a.elapsed_time(b) * 1e3 / 200:6.1f} us", flush=True)
