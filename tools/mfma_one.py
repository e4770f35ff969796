"""Runs every MFMA kernel form of the 40x256 step a few times at the headline shape (for rocprofv3 --pmc runs):
forward conv (plain / BatchNorm+ReLU+bias input), data gradient (two-tensor input; plain / masked epilogue), weight gradient
(plain / fused input).  MFMA_ONE=<comma list of fwd,fwd2,dgrad,dgradm,dgradmg,wgrad,wgradf> selects forms (default: all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, C = int(os.environ.get("CB_B", 4096)), 256
dev = 'cuda'
A = lambda: torch.randn(B, 81, C, device=dev).bfloat16()
x, x2, yprev, dy = A(), A(), A(), A()
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
out, dyo = torch.empty_like(x), torch.empty_like(x)
rows = _lib.query("ka_conv3x3_sqpart_rows", B)
bsum = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
e1 = torch.empty(rows, C, device=dev); e2 = torch.empty(rows, C, device=dev)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1; g = torch.randn(B, C, device=dev) * 0.1
mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
k3 = torch.cat([torch.rand(C, device=dev) + 0.5, 0.1 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)])
twg = int(os.environ.get("MFMA_ONE_WGS", 0))
ns = _lib.query("ka_wgrad_splits", B, C, C, twg)
slab = torch.empty(ns * 9 * C * C, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
st = _lib.stream_ptr
gate_add = torch.stack([torch.sigmoid(torch.randn(B, C, device=dev)), 0.05 * torch.randn(B, C, device=dev)]).contiguous()
forms = {
    "fwd": lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st()),
    "fwd2": lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, g, 1, bsum, sq, B, C, C, 1, st()),
    "dgrad": lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st()),
    "dgradm": lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st()),
    "dgradmg": lambda: _lib.call("ka_conv3x3_dgrad_fused_gated", x, gate_add, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st()),
    "wgrad": lambda: _lib.call("ka_conv3x3_wgrad", dy, x, None, None, None, 0, slab, dw, B, C, C, C, 0, twg, 1, st()),
    "wgradf": lambda: _lib.call("ka_conv3x3_wgrad", dy, x, sc, sh, g, 1, slab, dw, B, C, C, C, 0, twg, 1, st()),
}
sel = os.environ.get("MFMA_ONE", ",".join(forms)).split(",")
n = int(os.environ.get("MFMA_ONE_N", 5))
timing = os.environ.get("MFMA_ONE_TIME") == "1"
for name in sel:
    fn = forms[name]
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    if timing:
        ms = a.elapsed_time(b) / n
        print(f"{name:8s} {ms * 1e3:8.1f} us  {2.0 * B * 81 * 9 * C * C / ms / 1e9:7.0f} TFLOP/s", flush=True)
