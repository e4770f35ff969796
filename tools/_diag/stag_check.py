"""round 4: the staggered producer/consumer conv (KA_CONV_P_STAG=1) against the plain schedule -- bit identity of every output and
stand-alone times, forms 0 (plain input), 1 (transform input), 3 (two-tensor input, KA_CONV_P=2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keisei_amd import _lib
DEV = "cuda"; C = 256
def run_all(B, time_n=0):
    g = torch.Generator(device=DEV).manual_seed(B)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, x2 = (rnd(B, 81, C).to(torch.bfloat16) for _ in range(2))
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    gb = rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    def launch(kind, out, dyo, bsum, sq):
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
    def run(kind):
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        out, dyo = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16)
        bsum, sq = nan(B, C), nan(B, C)
        launch(kind, out, dyo, bsum, sq)
        torch.cuda.synchronize()
        ms = 0.0
        if time_n:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(time_n): launch(kind, out, dyo, bsum, sq)
            b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / time_n
        return (out, dyo, bsum, sq), ms
    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    os.environ["KA_CONV_P"] = "2"
    for rep in range(3 if time_n else 1):
        for kind in (0, 1, 3):
            os.environ["KA_CONV_P_STAG"] = "0"; _lib.reload_options(); ref, t0 = run(kind)
            os.environ["KA_CONV_P_STAG"] = "1"; _lib.reload_options(); got, t1 = run(kind)
            ok = all(same(a.float(), b.float()) for a, b in zip(ref, got)) and not bool(ref[0].float().isnan().any())
            print(f"B={B} kind={kind} identical={ok} plain {t0 * 1e3:.1f} us staggered {t1 * 1e3:.1f} us", flush=True)
            assert ok
run_all(515); run_all(1024); run_all(4096, time_n=30)
