"""Stand-alone timing of the HBM-bound board kernels at the headline shape (B=4096, C=256, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
B, C, H = 4096, 256, 16
dt, code, dev = torch.bfloat16, 1, 'cuda'
A = lambda: torch.randn(B, 81, C, device=dev).to(dt)
dout, out, y, x, dz, dxc = A(), A(), A(), A(), A(), A()
sc = torch.rand(C, device=dev) + 0.5; sh = 0.1 * torch.randn(C, device=dev); mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
se = torch.randn(B, 2 * C, device=dev); se1 = torch.randn(B, H, device=dev)
W2 = torch.randn(2 * C, H, device=dev) / 4; W1 = torch.randn(H, C, device=dev) / 16
dse = torch.empty(B, 2 * C, device=dev); dh = torch.empty(B, H, device=dev); dsq = torch.randn(B, C, device=dev)
s1 = torch.empty(B, C, device=dev); s2 = torch.empty(B, C, device=dev)
pool = torch.empty(B, 4 * C, device=dev); dpool = torch.randn(B, 3 * C, device=dev)
st = _lib.stream_ptr
abytes = B * 81 * C * 2
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
_lib.call("ka_pool_fwd", x, pool, B, C, code, st()); 
cases = [
    ("tail_bwd_fused (3R+1W)", 4, lambda: _lib.call("ka_tail_bwd_fused", dout, out, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2, B, C, H, code, st())),
    ("tail_bwd_reduce (3R)", 3, lambda: _lib.call("ka_tail_bwd_reduce", dout, out, y, sc, sh, se, dse, B, C, code, st())),
    ("tail_bwd_dz (3R+1W)", 4, lambda: _lib.call("ka_tail_bwd_dz", dout, out, y, se, dsq, mu, istd, dz, s1, s2, B, C, code, st())),
    ("block_tail_fwd (2R+1W)", 3, lambda: _lib.call("ka_block_tail_fwd", y, sc, sh, se, x, out, pool, B, C, code, st())),
    ("block_dx (4R+1W)", 5, lambda: _lib.call("ka_block_dx", dxc, dout, out, x, pool, dpool, dz, B, C, code, st())),
]
dx2 = torch.empty_like(dz)
cases += [
    ("block_dx + tail_bwd, two launches (7R+2W)", 9, lambda: (_lib.call("ka_block_dx", dxc, dout, out, x, pool, dpool, dx2, B, C, code, st()),
                                                              _lib.call("ka_tail_bwd_fused", dx2, x, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2, B, C, H, code, st()))),
    ("block_dx_tail_bwd, one launch (5R+2W)", 7, lambda: _lib.call("ka_block_dx_tail_bwd", dxc, dout, out, x, pool, dpool, dx2, y, sc, sh, se, se1, W2, W1, mu, istd,
                                                                   dz, dse, dh, s1, s2, B, C, H, code, st())),
    ("block_dx_tail_bwd_du, chain form (4R+2W)", 6, lambda: _lib.call("ka_block_dx_tail_bwd_du", dxc, dout, x, pool, dpool, dx2, y, sc, sh, se, se1, W2, W1, mu, istd,
                                                                      dz, dse, dh, s1, s2, B, C, H, code, st())),
]
gate_add = torch.empty(2, B, C, device=dev)
def gate_form(wpe):
    def fn():
        _lib.call("ka_block_dx_tail_bwd_du_gate", dxc, dout, x, pool, dpool, dx2, y, sc, sh, se, se1, W2, W1, mu, istd,
                  gate_add[0], gate_add[1], dse, dh, s1, s2, B, C, H, code, st())
    return fn
cases += [("block_dx_tail_bwd_du_gate, no dz (4R+1W)", 5, gate_form(4))]
for name, passes, fn in cases:
    ms = timeit(fn)
    print(f"{name:44s} {ms * 1e3:8.1f} us  {passes * abytes / ms / 1e9:6.2f} TB/s", flush=True)
for p4 in (1, 0, 1):
    os.environ["KA_TAIL_GATE_P4"] = str(p4); _lib.reload_options()
    ms = timeit(gate_form(4))
    print(f"{'block_dx_tail_bwd_du_gate, KA_TAIL_GATE_P4=' + str(p4):44s} {ms * 1e3:8.1f} us  {5 * abytes / ms / 1e9:6.2f} TB/s", flush=True)
os.environ.pop("KA_TAIL_GATE_P4"); _lib.reload_options()
for kb in (0, 6):
    os.environ["KA_TAIL_FWD_KB"] = str(kb); _lib.reload_options()
    ms = timeit(lambda: _lib.call("ka_block_tail_fwd", y, sc, sh, se, x, out, pool, B, C, code, st()))
    print(f"{'block_tail_fwd (2R+1W), KA_TAIL_FWD_KB=' + str(kb):44s} {ms * 1e3:8.1f} us  {3 * abytes / ms / 1e9:6.2f} TB/s", flush=True)
