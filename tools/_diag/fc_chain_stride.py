"""Does the row stride of W1 matter to fc_chain_kernel?  The 768-128-256 chain with K1 = 768 / 784 / 800 / 832 / 1024 (the extra inputs are
zeros: same result, 0-33 % more work), cold (512 MB fill before each launch) and warm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
dev, M, H, N2 = "cuda", 4096, 128, 256
st = _lib.stream_ptr
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for K1 in (768, 784, 800, 832, 1024):
    x, W1, b1, W2, b2 = torch.randn(M, K1, device=dev), torch.randn(H, K1, device=dev) / K1 ** 0.5, torch.randn(H, device=dev), torch.randn(N2, H, device=dev) / H ** 0.5, torch.randn(N2, device=dev)
    hid, y = torch.empty(M, H, device=dev), torch.empty(M, N2, device=dev)
    fn = lambda: _lib.call("ka_fc_chain", x, None, None, 1.0, W1, b1, W2, b2, None, hid, y, M, K1, K1, H, N2, st())
    if not _lib.query("ka_fc_chain_supported", K1, K1, H, N2):
        print(K1, "unsupported"); continue
    for _ in range(5): fn()
    res = {}
    for mode in ("cold", "warm"):
        ts = []
        for it in range(40):
            if mode == "cold": big.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort(); res[mode] = ts[len(ts) // 2]
    print(f"K1 = {K1:5d} (row stride {K1 * 4} B): cold {res['cold']:.1f} us  warm {res['warm']:.1f} us", flush=True)
