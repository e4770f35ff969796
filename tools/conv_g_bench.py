"""An opt-in conv kernel against conv3x3_kernel on the four launch kinds: outputs, statistics, stand-alone time.
CG_VAR names the switch (KA_CONV_G: the GEMM-class kernel, KA_CONV_T: the streaming form); =0 selects conv3x3_kernel
inside the same entry points.  CG_KINDS = launch kinds (0 plain fwd, 1 transformed fwd, 2 masked dgrad, 3 plain dgrad)."""
import os, sys
sys.path.insert(0, '.')
import torch
from keisei_amd import _lib
dev = 'cuda'
VAR = os.environ.get('CG_VAR', 'KA_CONV_G')
ON = os.environ.get('CG_ON', '1')            # the switch's "on" value (KA_CONV_P=3 routes every launch kind)
C = 256
dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
sizes = [int(v) for v in (sys.argv[1:] or ["770", "1000", "3840", "4096"])]
for B in sizes:
    x = torch.randn(B, 81, C, device=dev).to(dt); x2 = torch.randn(B, 81, C, device=dev).to(dt); yprev = torch.randn(B, 81, C, device=dev).to(dt)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1; g = torch.randn(B, C, device=dev) * 0.1
    k3 = torch.cat([torch.rand(C, device=dev) + 0.5, 0.1 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)])
    mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
    def run(kind, outs):
        out, bsum, sq, dyo, e1, e2 = outs
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 4: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, None, None, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, g, 1, bsum, sq, B, C, C, 1, st)
        if kind == 2: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
    names = ["plain (conv1 fwd)", "bn+relu+bias input (conv2 fwd)", "dgrad fused, masked epilogue (conv2 bwd)", "dgrad fused, plain epilogue (conv1 bwd)", "plain, no statistics"]
    for kind in (int(k) for k in os.environ.get('CG_KINDS', '0,1').split(',')):
        res = {}
        for flag in ("0", "1"):
            os.environ[VAR] = ON if flag == "1" else "0"
            outs = [torch.full((B, 81, C), float("nan"), device=dev).to(dt), torch.full((B, C), float("nan"), device=dev), torch.full((B, C), float("nan"), device=dev),
                    torch.full((B, 81, C), float("nan"), device=dev).to(dt), torch.full((B, C), float("nan"), device=dev), torch.full((B, C), float("nan"), device=dev)]
            run(kind, outs); torch.cuda.synchronize()
            res[flag] = [outs, 1e9]
        for rnd in range(3):                      # alternating rounds in one process, best of three
            for flag in ("0", "1"):
                os.environ[VAR] = ON if flag == "1" else "0"
                res[flag][1] = min(res[flag][1], timeit(lambda: run(kind, res[flag][0])))
        o0, t0 = res["0"]; o1, t1 = res["1"]
        def rel(a, b):
            a, b = a.float(), b.float()
            if torch.isnan(a).all() and torch.isnan(b).all(): return 0.0
            return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()
        diffs = [rel(o1[i], o0[i]) for i in range(6)]
        flop = 2.0 * B * 81 * 9 * C * C
        print(f"B={B} {names[kind]:42s}: conv3x3 {t0:.4f} ms  {VAR} {t1:.4f} ms ({flop / t1 / 1e9:.0f} TF)  rel diffs out/bsum/sq|s1/dy/s1|-/s2 "
              + " ".join(f"{d:.1e}" for d in diffs), flush=True)
