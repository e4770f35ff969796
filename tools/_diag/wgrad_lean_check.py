"""round 4: wgrad_flat_kernel (KA_WGRAD_LEAN=1, default) against wgrad_kernel<bf16_t, 128> -- bit identity of dW and stand-alone times,
plain and fused (BatchNorm + ReLU + bias) input, the tower shape, a ragged batch and the stem shape (Cin 64 padded, 50 real)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
DEV = "cuda"
for B, Cin, Cin_real, Cout, n in ((515, 256, 256, 256, 0), (700, 64, 50, 256, 0), (130, 128, 128, 128, 0), (4096, 256, 256, 256, 30)):
    g = torch.Generator(device=DEV).manual_seed(B)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    dy, x = rnd(B, 81, Cout).bfloat16(), rnd(B, 81, Cin).bfloat16()
    sc, sh, gb = torch.rand(Cin, device=DEV, generator=g) + 0.5, rnd(Cin) * 0.1, rnd(B, Cin) * 0.1
    ns = _lib.query("ka_wgrad_splits", B, Cin, Cout, 0)
    slab = torch.empty(ns * 9 * Cout * Cin, device=DEV)
    def run(fused, time_n):
        dw = torch.full((Cout, Cin_real, 3, 3), float("nan"), device=DEV)
        args = (sc, sh, gb, 1) if fused else (None, None, None, 0)
        fn = lambda: _lib.call("ka_conv3x3_wgrad", dy, x, *args, slab, dw, B, Cin, Cin_real, Cout, 0, 0, 1, _lib.stream_ptr())
        fn(); torch.cuda.synchronize()
        ms = 0.0
        if time_n:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(time_n): fn()
            b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / time_n
        return dw, ms
    for rep in range(3 if n else 1):
        for fused in (False, True):
            os.environ["KA_WGRAD_LEAN"] = "0"; _lib.reload_options(); ref, t0 = run(fused, n)
            os.environ["KA_WGRAD_LEAN"] = "1"; _lib.reload_options(); got, t1 = run(fused, n)
            ok = bool((ref == got).all()) and not bool(ref.isnan().any())
            print(f"B={B} Cin={Cin} Cout={Cout} fused={fused} identical={ok} wgrad_kernel {t0 * 1e3:.1f} us lean {t1 * 1e3:.1f} us", flush=True)
            assert ok, float((ref - got).abs().max())
