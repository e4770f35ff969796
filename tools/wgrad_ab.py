"""Weight-gradient kernel forms against each other in one process (B = 4096 and ragged B, C = 256, bf16): the first form
(default) vs KA_WGRAD_V=<WG_AB_V, default 2> -- 2: the software-pipelined flat-K form, bit-identical; 3: its 4-wave "lite"
form with a 128 x 32 slab (another split count: equal to fp32 rounding) -- and alternating timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
C, dev = 256, 'cuda'
def run(B, fused, v, dy, x, sc, sh, g, twg=0):
    if v: os.environ.pop("KA_WGRAD_V", None)
    else: os.environ["KA_WGRAD_V"] = os.environ.get("WG_AB_V", "2")
    ns = _lib.query("ka_wgrad_splits", B, C, C, twg)
    slab = torch.zeros(ns * 9 * C * C, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
    args = (sc, sh, g, 1) if fused else (None, None, None, 0)
    fn = lambda: _lib.call("ka_conv3x3_wgrad", dy, x, *args, slab, dw, B, C, C, C, 0, twg, 1, _lib.stream_ptr())
    fn(); torch.cuda.synchronize()
    return fn, dw, slab
for B in (int(os.environ.get("CB_B", 4096)), 777, 5):
    gen = torch.Generator(device=dev).manual_seed(B)
    x = torch.randn(B, 81, C, device=dev, generator=gen).bfloat16(); dy = torch.randn(B, 81, C, device=dev, generator=gen).bfloat16()
    sc = torch.rand(C, device=dev, generator=gen) + 0.5; sh = torch.randn(C, device=dev, generator=gen) * 0.1
    g = torch.randn(B, C, device=dev, generator=gen) * 0.1
    for fused in (False, True):
        for twg in (0, 192):
            f_new, dw_new, sl_new = run(B, fused, 0, dy, x, sc, sh, g, twg)
            f_old, dw_old, sl_old = run(B, fused, 1, dy, x, sc, sh, g, twg)
            same = torch.equal(dw_new, dw_old) and torch.equal(sl_new, sl_old)
            md = float((dw_new - dw_old).abs().max())
            line = (f"B={B} fused={fused} target_wgs={twg}: bit-identical={same} max|diff|={md:.3e} (|dw|max {float(dw_old.abs().max()):.1f}) "
                    f"finite={bool(torch.isfinite(dw_new).all())}")
            if B >= 4096:
                best = {"new": 1e9, "old": 1e9}
                for rep in range(3):
                    for name, v in (("new", 0), ("old", 1)):
                        fn, _, _ = run(B, fused, v, dy, x, sc, sh, g, twg)
                        for _ in range(3): fn()
                        torch.cuda.synchronize()
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        for _ in range(20): fn()
                        b.record(); torch.cuda.synchronize()
                        best[name] = min(best[name], a.elapsed_time(b) / 20)
                line += f"  new {best['new'] * 1e3:.1f} us  old {best['old'] * 1e3:.1f} us (wgrad + reduce, best of 3 x 20)"
            print(line, flush=True)
