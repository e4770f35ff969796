"""Diagnostic: where does a conv3x3 workgroup spend its cycles? (s_memtime stamps at phase boundaries)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, C = 4096, 256
dt = torch.bfloat16; code = 1; dev = 'cuda'
x = torch.randn(B, 81, C, device=dev).to(dt)
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, code, _lib.stream_ptr())
out = torch.empty_like(x); rows = _lib.query("ka_conv3x3_sqpart_rows", B)
bsum = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
for kc, ntw, wm in [(128, 4, 2), (64, 4, 2)]:
    os.environ["KA_CONV_KC"] = str(kc); os.environ["KA_CONV_NTW"] = str(ntw); os.environ["KA_CONV_WM"] = "1"; _lib.reload_options()
    nwg = B
    import time
    t0 = time.time()
    while time.time() - t0 < 2.5:         # the clock the chip holds under this load settles over seconds of launches
        for _ in range(50):
            _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, code, _lib.stream_ptr())
        torch.cuda.synchronize()
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    _lib.call("ka_debug_conv_stamps", stamps)
    _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, code, _lib.stream_ptr())
    torch.cuda.synchronize()
    _lib.call("ka_debug_conv_stamps", None)
    s = stamps.cpu().view(nwg, 8).double()
    tot = (s[:, 7] - s[:, 0]).mean()
    real = (s[:, 4] - s[:, 3])
    ok = real > 0
    ghz = ((s[:, 7] - s[:, 0])[ok] / real[ok] * 0.1).median()
    print(f"in-kernel clock (median over workgroups): {ghz:.3f} GHz -> dense bf16 MFMA rate at that clock {2500 * ghz / 2.4:.0f} TFLOP/s")
    print(f"KC={kc} NTW={ntw}: stage0 {(s[:, 1] - s[:, 0]).mean():.0f}  chunks {(s[:, 2] - s[:, 1]).mean():.0f}  epilogue(wave 0) {(s[:, 7] - s[:, 2]).mean():.0f} | total {tot:.0f}")
