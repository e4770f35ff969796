"""Does an HBM-bound board kernel run at its stand-alone rate beside an MFMA-bound kernel ON THE SAME CUs (co-residency), and
what does the MFMA kernel lose?  The MFMA side is a synthetic register-resident burn (tools/_diag/mfma_burn.hip, compiled here
with hipcc) with ~300 of the 512 registers per SIMD lane and a chosen LDS footprint -- the footprint a weight-gradient kernel
would need to leave room for ka_tail_bwd_fused / ka_block_dx (<= 80 VGPRs x 512 threads, <= 32 KB LDS).
    python tools/_diag/coresidency.py [lds_kb ...]"""
import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), "mfma_burn.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "mfma_burn.hip")], check=True)
burn = ctypes.CDLL(so).mfma_burn
burn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B, C, H, dev = 4096, 256, 16, 'cuda'
A = lambda: torch.randn(B, 81, C, device=dev).bfloat16()
dout, out, y, x, dz, dxc = A(), A(), A(), A(), A(), A()
sc = torch.rand(C, device=dev) + 0.5; sh = 0.1 * torch.randn(C, device=dev); mu = 0.1 * torch.randn(C, device=dev); istd = torch.rand(C, device=dev) + 0.5
se = torch.randn(B, 2 * C, device=dev); se1 = torch.randn(B, H, device=dev)
W2 = torch.randn(2 * C, H, device=dev) / 4; W1 = torch.randn(H, C, device=dev) / 16
dse = torch.empty(B, 2 * C, device=dev); dh = torch.empty(B, H, device=dev)
s1 = torch.empty(B, C, device=dev); s2 = torch.empty(B, C, device=dev)
pool = torch.empty(B, 4 * C, device=dev); dpool = torch.randn(B, 3 * C, device=dev)
_lib.call("ka_pool_fwd", x, pool, B, C, 1, _lib.stream_ptr())
sink = torch.empty(256 * 512, device=dev)
side = torch.cuda.Stream()
def tail(st): _lib.call("ka_tail_bwd_fused", dout, out, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2, B, C, H, 1, st)
def dx(st): _lib.call("ka_block_dx", dxc, dout, out, x, pool, dpool, dz, B, C, 1, st)
def wall(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
main = torch.cuda.current_stream()
for lds_kb in [int(v) for v in sys.argv[1:]] or [96, 124, 150]:
    iters = 2600                                         # ~ 0.4 ms of MFMAs per workgroup
    def burn_only(): burn(sink.data_ptr(), 256, iters, lds_kb * 1024, main.cuda_stream)
    def board_only(): tail(main.cuda_stream); dx(main.cuda_stream)
    def both():
        side.wait_stream(main)
        burn(sink.data_ptr(), 256, iters, lds_kb * 1024, side.cuda_stream)
        tail(main.cuda_stream); dx(main.cuda_stream)
        main.wait_stream(side)
    def both_burn_first():                               # the MFMA kernel already owns the CUs when the board kernels arrive
        side.wait_stream(main)
        burn(sink.data_ptr(), 256, iters, lds_kb * 1024, side.cuda_stream)
        torch.cuda._sleep(40000)                         # ~ 20 us on the main stream before the board kernels are queued
        tail(main.cuda_stream); dx(main.cuda_stream)
        main.wait_stream(side)
    tb, tk, t2, t3 = wall(burn_only), wall(board_only), wall(both), wall(both_burn_first)
    print(f"burn LDS {lds_kb:3d} KB: burn alone {tb:7.1f} us | tail_bwd + block_dx alone {tk:7.1f} us | together {t2:7.1f} us "
          f"(burn first: {t3:7.1f}) | serial sum {tb + tk:7.1f}  ideal overlap {max(tb, tk):7.1f}", flush=True)
