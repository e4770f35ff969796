"""Register-resident MFMA burn, 16x16x32 vs 32x32x16 (bf16): sustained rate of each instruction shape with no operand traffic
at all -- the ceiling the DVFS-held clock leaves, and whether the larger shape (half the operand-register reads per FLOP)
holds a higher clock.  Run on an MI355X from the repository root."""
import ctypes, os, subprocess, tempfile
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), "mfma_burn.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "mfma_burn.hip")], check=True)
lib = ctypes.CDLL(so)
for f in (lib.mfma_burn, lib.mfma_burn32): f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.empty(256 * 512, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 2600
flops = 256 * 8 * iters * 32 * 16 * 16 * 32 * 2
for rep in range(2):
    for name, fn in (("16x16x32", lib.mfma_burn), ("32x32x16", lib.mfma_burn32)):
        for _ in range(2): fn(sink.data_ptr(), 256, iters, 4096, st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn(sink.data_ptr(), 256, iters, 4096, st)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        # 8 waves per CU = 2 per SIMD, iters x 512 matrix-pipe cycles per wave -> cycles per SIMD; held clock = cycles / time
        cyc = 2 * iters * 32 * 16
        print(f"{name}: {ms * 1e3:8.1f} us  {flops / ms / 1e12:6.2f} PFLOP/s  held clock if the pipe never idles {cyc / ms / 1e6:5.2f} GHz", flush=True)
