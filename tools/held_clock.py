"""The shader clock each tower MFMA kernel holds (MI355X_MICROARCH.md, "DVFS give-back" item 6): after >= 2 s of back-to-back
launches of the form on random data, one launch with the stamp buffer set (ka_debug_conv_stamps): every workgroup records
s_memtime / s_memrealtime around its main loop; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.
    python tools/held_clock.py [tag]  ->  gpurun_out/<tag>_held_clock.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
B, C = 4096, 256
dev = "cuda"
A = lambda: torch.randn(B, 81, C, device=dev).bfloat16()
x, x2, yprev, dy = A(), A(), A(), A()
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
out, dyo = torch.empty_like(x), torch.empty_like(x)
bsum = torch.empty(B, C, device=dev); sq = torch.empty(B, C, device=dev); e1 = torch.empty(B, C, device=dev); e2 = torch.empty(B, C, device=dev)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1; g = torch.randn(B, C, device=dev) * 0.1
mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
k3 = torch.cat([torch.rand(C, device=dev) + 0.5, 0.1 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)])
ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
slab = torch.empty(ns * 9 * C * C, device=dev); dw = torch.empty(C, C, 3, 3, device=dev)
st = _lib.stream_ptr
gate_add = torch.stack([torch.sigmoid(torch.randn(B, C, device=dev)), 0.05 * torch.randn(B, C, device=dev)]).contiguous()
forms = {
    "conv3x3_pc2_kernel, plain input (conv1 forward)": lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st()),
    "conv3x3_pc2_kernel, transform input (conv2 forward)": lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, g, 1, bsum, sq, B, C, C, 1, st()),
    "conv3x3_pc2_kernel, two-tensor input (conv1 data gradient)": lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st()),
    "conv3x3_pc2_kernel, two-tensor input + masked epilogue (conv2 data gradient)": lambda: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st()),
    "conv3x3_pc2_kernel, GATED two-tensor input + masked epilogue (conv2 data gradient as shipped)": lambda: _lib.call("ka_conv3x3_dgrad_fused_gated", x, gate_add, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st()),
    "wgrad_flat_kernel, plain input (conv1 weight gradient)": lambda: _lib.call("ka_conv3x3_wgrad", dy, x, None, None, None, 0, slab, dw, B, C, C, C, 0, 0, 1, st()),
    "wgrad_flat_kernel, fused input (conv2 weight gradient)": lambda: _lib.call("ka_conv3x3_wgrad", dy, x, sc, sh, g, 1, slab, dw, B, C, C, C, 0, 0, 1, st()),
}
lines = [f"device: {torch.cuda.get_device_name(0)}; B = {B}, C = {C}, bf16, random data; 2.5 s of back-to-back launches per form, then one stamped launch",
         "clock = median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz around the workgroup's main loop (corner launch excluded)"]
for name, fn in forms.items():
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    _lib.call("ka_debug_conv_stamps", stamps)
    fn(); torch.cuda.synchronize()
    _lib.call("ka_debug_conv_stamps", None)
    s = stamps.cpu().view(-1, 8).double()
    ok = (s[:, 4] - s[:, 3]) > 0
    ghz = ((s[:, 7] - s[:, 0])[ok] / (s[:, 4] - s[:, 3])[ok] * 0.1)
    cyc = (s[:, 7] - s[:, 0])[ok]
    lines.append(f"{name:96s} {us:7.1f} us/launch (incl. corner)  clock {float(ghz.median()):.3f} GHz (p10 {float(ghz.quantile(0.1)):.3f}, p90 {float(ghz.quantile(0.9)):.3f})  "
                 f"main loop {float(cyc.median()) / 1e3:.0f} k cycles, {int(ok.sum())} workgroups")
    print(lines[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
open(f"gpurun_out/{tag}_held_clock.txt", "w").write("\n".join(lines) + "\n")
