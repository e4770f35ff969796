"""Register-resident attention forward / backward at the benched shape (4096 boards, 8 heads of 32), with and without dropout.
python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, H, dh, dev = int(os.environ.get("AB_B", 4096)), 8, 32, 'cuda'
d = H * dh
st = _lib.stream_ptr
qkv = torch.randn(B * 81, 3 * d, device=dev).bfloat16()
out = torch.empty(B * 81, d, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H * 81, device=dev)
dout = torch.randn(B * 81, d, device=dev).bfloat16(); dqkv = torch.empty_like(qkv)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for p in (0.0, 0.1):
    f = timeit(lambda: _lib.call("ka_tf_attention_fwd", qkv, out, lse, B, H, dh, p, 12345, _lib.DTYPE_BF16, st()))
    b = timeit(lambda: _lib.call("ka_tf_attention_bwd", qkv, dout, lse, dqkv, B, H, dh, p, 12345, _lib.DTYPE_BF16, st()))
    b2 = timeit(lambda: _lib.call("ka_tf_attention_bwd_o", qkv, out, dout, lse, dqkv, B, H, dh, p, 12345, _lib.DTYPE_BF16, st()))
    print(f"dropout {p}: forward {f:7.1f} us   backward {b:7.1f} us   backward in two launches (dQ | dK,dV) {b2:7.1f} us", flush=True)
