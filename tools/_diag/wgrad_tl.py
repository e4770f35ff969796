"""round 4 diagnostic (needs the -DKA_DIAG_WGRAD_TL build of wgrad.hip, tools/_diag/build_variants.sh): where do the waves of one
wgrad_flat_kernel workgroup spend a board?  Per wave: stage (before the k-steps, early waves), k-steps, stage (after, late waves),
barrier wait -- s_memtime cycles, averaged over boards 8..23 of workgroup 0, after 1.5 s of back-to-back launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
DEV = "cuda"
B, C = 4096, 256
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
dy, x = rnd(B, 81, C).bfloat16(), rnd(B, 81, C).bfloat16()
sc, sh, gb = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1, rnd(B, C) * 0.1
ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
slab = torch.empty(ns * 9 * C * C, device=DEV)
dw = torch.empty(C, C, 3, 3, device=DEV)
for stag, fused in ((1, False), (0, False), (1, True), (0, True)):
    os.environ["KA_WGRAD_STAG"] = str(stag); _lib.reload_options()
    args = (sc, sh, gb, 1) if fused else (None, None, None, 0)
    fn = lambda: _lib.call("ka_conv3x3_wgrad", dy, x, *args, slab, dw, B, C, C, C, 0, 0, 1, _lib.stream_ptr())
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    stamps = torch.zeros(4096 * 8 + 16 * 8 * 8, dtype=torch.int64, device=DEV)
    _lib.call("ka_debug_conv_stamps", stamps)
    fn(); torch.cuda.synchronize()
    _lib.call("ka_debug_conv_stamps", None)
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(40): fn()
    b_.record(); torch.cuda.synchronize()
    print(f"(un-stamped launches, incl. the slab reduce: {a_.elapsed_time(b_) / 40 * 1e3:.1f} us)")
    s = stamps.cpu()[4096 * 8:].view(16, 8, 8).double()          # [board][wave][phase]
    print(f"fused={fused} KA_WGRAD_STAG={stag}: cycles per board (mean over boards 8..23 of workgroup 0); board period = {float((s[1:, :, 0] - s[:-1, :, 0]).mean()):.0f}")
    print("wave  stage-before   k-steps  stage-after  barrier-wait   (min..max k-steps)")
    for w in range(8):
        d = [(s[:, w, i + 1] - s[:, w, i]) for i in range(4)]
        print(f"{w:4d} {float(d[0].mean()):12.0f} {float(d[1].mean()):10.0f} {float(d[2].mean()):12.0f} {float(d[3].mean()):13.0f}   ({float(d[1].min()):.0f}..{float(d[1].max()):.0f})")
    # when, inside the board period, does each wave run its k-steps (relative to wave 0's loop top)
    t_ref = s[:, 0:1, 0]
    print("k-step windows relative to wave 0's loop top (mean start .. mean end):",
          "  ".join(f"w{w}:{float((s[:, w, 1] - t_ref[:, 0]).mean()):.0f}..{float((s[:, w, 2] - t_ref[:, 0]).mean()):.0f}" for w in range(8)))
