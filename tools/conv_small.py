"""conv3x3 latency at rollout batch sizes for the n-tiles-per-wave choices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
C = 256; dt = torch.bfloat16; code = 1; dev = 'cuda'
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, code, _lib.stream_ptr())
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B in (64, 128, 256, 512):
    x = torch.randn(B, 81, C, device=dev).to(dt); out = torch.empty_like(x); g = torch.randn(B, C, device=dev)
    bsum = torch.empty(B, C, device=dev)
    for ntw in ("4", "2", "1"):
        os.environ["KA_CONV_NTW"] = ntw; _lib.reload_options()
        ms = timeit(lambda: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, g, 1, bsum, None, B, C, C, code, _lib.stream_ptr()))
        print(f"B={B} NTW={ntw}: {ms * 1e3:.1f} us", flush=True)
