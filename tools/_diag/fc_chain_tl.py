"""round 4 diagnostic (needs the -DKA_DIAG_FC_TL build of gemm.hip): where do the waves of a fc_chain_kernel workgroup spend a launch?
s_memtime per wave at: start, x' staged, past the barrier, phase 1 done, past the barrier, hidden written, end.  Shapes of the 40x256 step."""
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
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for name, fn in (("gpool fwd 768-128-256", fwd(768, 128, 256, False)), ("se fwd 256-16-512 (affine)", fwd(256, 16, 512, True)),
                 ("gpool bwd 256-128-768", bwd(256, 128, 768))):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    acc = torch.zeros(3, 8, 7, dtype=torch.float64)
    n = 10
    for it in range(n):
        big.zero_()                                  # (the step's kernels between two chain launches leave none of its operands in the L2)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        _lib.call("ka_debug_conv_stamps", stamps)
        fn(); torch.cuda.synchronize()
        _lib.call("ka_debug_conv_stamps", None)
        acc += stamps.cpu()[:3 * 8 * 7 + 3 * 8].view(-1)[: 3 * 8 * 8].view(3, 8, 8)[:, :, :7].double()
    s = acc / n
    t0 = s[:, :, 0].min(dim=1, keepdim=True).values
    print(name)
    for wg in range(3):
        d = s[wg] - t0[wg]
        print(f"  workgroup {wg * 100}: phase ends (cycles after the first wave's start, mean / max over waves):  " +
              "  ".join(f"{lbl} {float(d[:, i].mean()):.0f}/{float(d[:, i].max()):.0f}" for i, lbl in
                        enumerate(("start", "staged", "barrier", "phase1", "barrier", "hidden", "end"))))
