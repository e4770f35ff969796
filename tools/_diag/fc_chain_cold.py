"""fc_chain_kernel with its operands NOT in the L2 (as in the step, where 170 MB tensors pass between two launches of a layer):
a 256 MB fill before every timed launch; median of 40."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
dev, M = "cuda", 4096
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
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name, fn in (("gpool fwd 768-128-256", fwd(768, 128, 256, False)), ("se fwd 256-16-512 (affine)", fwd(256, 16, 512, True)),
                 ("gpool bwd 256-128-768", bwd(256, 128, 768))):
    for _ in range(5): fn()
    ts = []
    for it in range(40):
        big.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{name:28s} cold median {ts[len(ts) // 2]:6.1f} us  (min {ts[0]:.1f})", flush=True)
