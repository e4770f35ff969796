"""Stand-alone time of ka_pool_fwd (pooled statistics of a stored block output) at the headline shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
B, C = 4096, 256
x = torch.relu(torch.randn(B, 81, C, device="cuda")).bfloat16()
pool = torch.empty(B, 4 * C, device="cuda")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ms = t(lambda: _lib.call("ka_pool_fwd", x, pool, B, C, 1, _lib.stream_ptr()))
print(f"ka_pool_fwd (1R): {ms * 1e3:.1f} us  {B * 81 * C * 2 / ms / 1e9:.2f} TB/s")
