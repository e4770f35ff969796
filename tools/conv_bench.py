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
    os.environ["KA_CONV_KC"] = "128"; os.environ["KA_CONV_NTW"] = "4"; os.environ["KA_CONV_WM"] = "2"
    for rep in range(2):
        for stg, prio in [(0, 0), (1, 0), (2, 0), (3, 0), (5, 0), (0, 1), (2, 1), (3, 1)]:
            os.environ["KA_CONV_STAGGER"] = str(stg); os.environ["KA_CONV_PRIO"] = str(prio)
            ms = timeit(lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, code, _lib.stream_ptr()), n=20)
            print(f"stagger={stg} prio={prio}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
elif which == "conv":
    for kc, ntw, wm in [(128, 4, 2), (64, 4, 2), (128, 4, 1), (64, 4, 1), (256, 4, 1)]:
        os.environ["KA_CONV_KC"] = str(kc); os.environ["KA_CONV_NTW"] = str(ntw); os.environ["KA_CONV_WM"] = str(wm)
        for name, args in (("plain", (None, None, None, 0)), ("fused", (sc, sh, g, 1))):
            ms = timeit(lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, args[0], args[1], args[2], args[3], bsum, sq, B, C, C, code, _lib.stream_ptr()))
            print(f"conv KC={kc:3d} NTW={ntw} WM={wm} {name}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
else:
    ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
    slab = torch.empty(ns * 9 * C * C, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
    dy = torch.randn(B, 81, C, device=dev).to(dt)
    for name, args in (("plain", (None, None, None, 0)), ("fused", (sc, sh, g, 1))):
        ms = timeit(lambda: _lib.call("ka_conv3x3_wgrad", dy, x, args[0], args[1], args[2], args[3], slab, dw, B, C, C, C, 0, 0, code, _lib.stream_ptr()))
        print(f"wgrad {name}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
