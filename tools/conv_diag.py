import ctypes, sys, os
sys.path.insert(0, '.')
import torch
B, C = 4096, 256
dev = 'cuda'
x = torch.randn(B, 81, C, device=dev).bfloat16()
w = torch.randn(C, C, 3, 3, device=dev) / 48
wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=dev)
out = torch.empty_like(x); rows = (B + 1) // 2
bsum = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
P = ctypes.c_void_p; I = ctypes.c_int
for name in ("keisei_amd/libkeisei_amd.so", "tools/_diag/libdiag_NO_W.so", "tools/_diag/libdiag_NO_A.so"):
    lib = ctypes.CDLL(name)
    lib.ka_pack_conv3x3.argtypes = [P, P, I, I, I, I, I, I, P]
    lib.ka_conv3x3_fwd.argtypes = [P] * 6 + [I, P, P, I, I, I, I, P]
    lib.ka_debug_conv_stamps.argtypes = [P]
    st = torch.cuda.current_stream().cuda_stream
    lib.ka_pack_conv3x3(w.data_ptr(), wp.data_ptr(), C, C, C, C, 0, 1, st)
    run = lambda: lib.ka_conv3x3_fwd(x.data_ptr(), wp.data_ptr(), out.data_ptr(), None, None, None, 0, bsum.data_ptr(), sq.data_ptr(), B, C, C, 1, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    stamps = torch.zeros(rows * 4, dtype=torch.int64, device=dev)
    lib.ka_debug_conv_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize(); lib.ka_debug_conv_stamps(None)
    s = stamps.cpu().view(rows, 4).double(); d = s[:, 1:] - s[:, :-1]
    print(f"{name:32s} {a.elapsed_time(b)/10:.4f} ms  stage {d[:,0].mean():.0f} main {d[:,1].mean():.0f} epi {d[:,2].mean():.0f}")
