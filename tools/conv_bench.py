"""Micro-benchmark of the conv3x3 / wgrad kernels at the headline shape (B=4096, C=256, bf16)."""
import os, sys, itertools
sys.path.insert(0, '.')
import torch
from keisei_amd import _lib
B, C = int(os.environ.get("CB_B", 4096)), int(os.environ.get("CB_C", 256))
dt = torch.bfloat16; code = 1
dev = 'cuda'
x = torch.randn(B, 81, C, device=dev).to(dt)
w = torch.randn(C, C, 3, 3, device=dev) / 48
cpk = 32
wp = torch.empty(9 * (C // cpk) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, code, _lib.stream_ptr())
out = torch.empty_like(x)
rows = _lib.query("ka_conv3x3_sqpart_rows", B)
bsum = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1; g = torch.randn(B, C, device=dev) * 0.1
flop = 2.0 * B * 81 * 9 * C * C
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
if which == "tune":
    os.environ["KA_CONV_KC"] = "128"; os.environ["KA_CONV_NTW"] = "4"; os.environ["KA_CONV_WM"] = "2"; _lib.reload_options()
    for rep in range(2):
        for stg, prio in [(0, 0), (1, 0), (2, 0), (3, 0), (5, 0), (0, 1), (2, 1), (3, 1)]:
            os.environ["KA_CONV_STAGGER"] = str(stg); os.environ["KA_CONV_PRIO"] = str(prio); _lib.reload_options()
            ms = timeit(lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, code, _lib.stream_ptr()), n=20)
            print(f"stagger={stg} prio={prio}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
elif which == "conv":
    k3 = torch.cat([torch.rand(C, device=dev) + 0.5, 0.1 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)])
    x2 = torch.randn(B, 81, C, device=dev).to(dt); yprev = torch.randn(B, 81, C, device=dev).to(dt)
    dyo = torch.empty_like(x); e1 = torch.empty(rows, C, device=dev); e2 = torch.empty(rows, C, device=dev)
    mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
    gate_add = torch.stack([torch.sigmoid(torch.randn(B, C, device=dev)), 0.05 * torch.randn(B, C, device=dev)]).contiguous()
    for kc, wm in ([(128, 1)] if os.environ.get("CB_QUICK") else [(128, 1), (64, 1), (128, 2)]):
        os.environ["KA_CONV_KC"] = str(kc); os.environ["KA_CONV_WM"] = str(wm); _lib.reload_options()
        for name, fn in (
            ("plain (conv1 fwd)", lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, code, _lib.stream_ptr())),
            ("bn+relu+bias input (conv2 fwd)", lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, g, 1, bsum, sq, B, C, C, code, _lib.stream_ptr())),
            ("dgrad fused, masked epilogue (conv2 bwd)", lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, code, _lib.stream_ptr())),
            ("dgrad fused GATED, masked epilogue (conv2 bwd)", lambda: _lib.call("ka_conv3x3_dgrad_fused_gated", x, gate_add, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, code, _lib.stream_ptr())),
            ("dgrad fused, plain epilogue (conv1 bwd)", lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, code, _lib.stream_ptr())),
            ("dgrad fused, plain epilogue, no dy write-back", lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, None, wp, out, None, None, None, None, None, None, None, None, B, C, C, code, _lib.stream_ptr())),
        ):
            ms = timeit(fn, n=20)
            print(f"conv KC={kc:3d} WM={wm} {name:42s}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
else:
    dy = torch.randn(B, 81, C, device=dev).to(dt)
    for tn in ("128", "64"):
        os.environ["KA_WGRAD_TN"] = tn; _lib.reload_options()
        for twg in (0, 192):
            ns = _lib.query("ka_wgrad_splits", B, C, C, twg)
            slab = torch.empty(ns * 9 * C * C, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
            for name, args in (("plain", (None, None, None, 0)), ("fused", (sc, sh, g, 1))):
                ms = timeit(lambda: _lib.call("ka_conv3x3_wgrad", dy, x, args[0], args[1], args[2], args[3], slab, dw, B, C, C, C, 0, twg, code, _lib.stream_ptr()))
                print(f"wgrad TN={tn} target={twg} splits={ns} {name}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
