"""Reference point only (not a product path): what the vendor library (MIOpen through torch.nn.functional.conv2d)
achieves on the tower's convolution shape, forward / data gradient / weight gradient, bf16."""
import torch, time
import torch.nn.functional as F
dev = 'cuda'
B, C = 4096, 256
flop = 2.0 * B * 81 * 9 * C * C
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
torch.backends.cudnn.benchmark = True
for fmt_name, fmt in (("NHWC", torch.channels_last), ("NCHW", torch.contiguous_format)):
    x = torch.randn(B, C, 9, 9, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
    w = (torch.randn(C, C, 3, 3, device=dev, dtype=torch.bfloat16) / 48).contiguous(memory_format=fmt)
    dy = torch.randn(B, C, 9, 9, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
    t = timeit(lambda: F.conv2d(x, w, padding=1))
    print(f"{fmt_name} forward : {t:.3f} ms  {flop / t / 1e9:.0f} TFLOP/s", flush=True)
    t = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
    print(f"{fmt_name} dgrad   : {t:.3f} ms  {flop / t / 1e9:.0f} TFLOP/s", flush=True)
    t = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
    print(f"{fmt_name} wgrad   : {t:.3f} ms  {flop / t / 1e9:.0f} TFLOP/s", flush=True)
