"""round 4: weight-gradient kernel with waves 4-7 staging after their MFMAs (KA_WGRAD_STAG=1) against the lockstep order --
bit identity of dW and stand-alone times, plain and fused (BatchNorm + ReLU + bias) input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
DEV = "cuda"; C = 256
for B, n in ((515, 0), (4096, 30)):
    g = torch.Generator(device=DEV).manual_seed(B)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    dy, x = rnd(B, 81, C).bfloat16(), rnd(B, 81, C).bfloat16()
    sc, sh, gb = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1, rnd(B, C) * 0.1
    ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
    slab = torch.empty(ns * 9 * C * C, device=DEV)
    def run(fused, time_n):
        dw = torch.full((C, C, 3, 3), float("nan"), device=DEV)
        args = (sc, sh, gb, 1) if fused else (None, None, None, 0)
        fn = lambda: _lib.call("ka_conv3x3_wgrad", dy, x, *args, slab, dw, B, C, C, C, 0, 0, 1, _lib.stream_ptr())
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
            os.environ["KA_WGRAD_STAG"] = "0"; _lib.reload_options(); ref, t0 = run(fused, n)
            os.environ["KA_WGRAD_STAG"] = "1"; _lib.reload_options(); got, t1 = run(fused, n)
            ok = bool((ref == got).all()) and not bool(ref.isnan().any())
            print(f"B={B} fused={fused} identical={ok} lockstep {t0 * 1e3:.1f} us staggered {t1 * 1e3:.1f} us", flush=True)
            assert ok
