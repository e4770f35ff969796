"""The transformer's K = 256 NT GEMMs at the benched shape (M = 4096 * 81 tokens): the activation-stationary kernel against the
tiled one (KA_TF_K256=0), per epilogue form.  python tools/tf_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keisei_amd import _lib
M, K, dev = 4096 * 81, 256, 'cuda'
st = _lib.stream_ptr
a = torch.randn(M, K, device=dev).bfloat16()
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (1024, 768, 256):
    b = (torch.randn(N, K, device=dev) / 16).bfloat16(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); res = torch.randn(M, N, device=dev).bfloat16(); act = torch.randn(M, N, device=dev).bfloat16()
    forms = {
        "plain": lambda: _lib.call("ka_tf_gemm_nt", a, b, out, None, None, M, N, K, K, K, N, 1, 0, 1, 0.0, 0, st()),
        "bias": lambda: _lib.call("ka_tf_gemm_nt", a, b, out, bias, None, M, N, K, K, K, N, 1, 0, 1, 0.0, 0, st()),
        "bias+relu+dropout": lambda: _lib.call("ka_tf_gemm_nt", a, b, out, bias, None, M, N, K, K, K, N, 1, 1, 1, 0.1, 7, st()),
        "bias+dropout+residual": lambda: _lib.call("ka_tf_gemm_nt", a, b, out, bias, res, M, N, K, K, K, N, 1, 0, 1, 0.1, 7, st()),
        "masked (relu_act + dropout)": lambda: _lib.call("ka_tf_gemm_nt_masked", a, b, out, act, M, N, K, K, K, N, 0.1, 7, st()),
    }
    for name, fn in forms.items():
        os.environ.pop("KA_TF_K256", None); t_new = timeit(fn)
        os.environ["KA_TF_K256"] = "0"; _lib.reload_options(); t_old = timeit(fn); os.environ.pop("KA_TF_K256"); _lib.reload_options()
        byts = (M * K + M * N * (1 + ("residual" in name) + ("masked" in name))) * 2
        print(f"N={N:5d} {name:28s} k256 {t_new:7.1f} us ({byts / t_new / 1e6:5.2f} TB/s, {2.0 * M * N * K / t_new / 1e6:6.0f} TF)   tiled {t_old:7.1f} us", flush=True)
