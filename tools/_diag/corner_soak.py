"""Soak: in-kernel corner against the corner launch, fresh random data every iteration (B = 4096, the three forms that take it), every
output and per-board sum bit for bit.  The sums' read-back goes past the vector cache (sc1): this is the check that it always does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
DEV, B, C = "cuda", 4096, 256
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
w = rnd(C, C, 3, 3) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    x, x2 = rnd(B, 81, C).to(torch.bfloat16), rnd(B, 81, C).to(torch.bfloat16)
    sc, sh, gb = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1, rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    res = {}
    for mode in ("0", "1"):
        os.environ["KA_CONV_CORNER_IN"] = mode; _lib.reload_options()
        outs = []
        for kind in (0, 1, 3):
            nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
            out, dyo, bsum, sq = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16), nan(B, C), nan(B, C)
            st = _lib.stream_ptr()
            if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
            if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
            if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, None, None, None, None, None, None, None, B, C, C, 1, st)
            outs += [out, dyo, bsum, sq]
        torch.cuda.synchronize()
        res[mode] = outs
    for a, b in zip(res["0"], res["1"]):
        if not bool(((a == b) | (a.isnan() & b.isnan())).all()):
            bad += 1
print("iterations with a difference:", bad)
sys.exit(1 if bad else 0)
