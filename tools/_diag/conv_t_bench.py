import os, sys
sys.path.insert(0, '.')
import torch
from keisei_amd import _lib
dev='cuda'; B=4096; C=256; dt=torch.bfloat16
x = torch.randn(B, 81, C, device=dev).to(dt)
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
_lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
out = torch.empty_like(x); bsum = torch.empty(B, C, device=dev); sq = torch.empty(B, C, device=dev)
out0 = torch.empty_like(x)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
run = lambda o: _lib.call("ka_conv3x3_fwd", x, wp, o, None, None, None, 0, bsum, sq, B, C, C, 1, _lib.stream_ptr())
for rnd in range(3):
    for flag in ("0", "1"):
        os.environ["KA_CONV_T"] = flag
        print(f"KA_CONV_T={flag}: {timeit(lambda: run(out if flag == '1' else out0)) * 1e3:7.1f} us", flush=True)
print("identical:", torch.equal(out, out0))
