"""GPU parity tests of the individual HIP kernels, called through the C ABI (keisei_amd._lib.call)
and compared with plain fp32 PyTorch CPU math / the oracle / the golden fixtures.

fp32 kernels (exact-fp32 MFMA / FMA paths) are held to rtol=atol=1e-5-class bounds scaled by the
tensor magnitude; bf16 kernels are compared against the same fp32 math evaluated on bf16-rounded
operands, with a bound of a few bf16 ulps (2^-8 relative) of the output magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from keisei_amd import _lib
from oracle import keisei_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def st():
    return _lib.stream_ptr()


def to_nhwc(x, dt):       # (B,C,9,9) cpu f32 -> (B,81,C) cuda T
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], 81, x.shape[1]).contiguous().to(dt).to(DEV)


def from_nhwc(y):         # (B,81,C) cuda -> (B,C,9,9) cpu f32
    return y.float().cpu().reshape(y.shape[0], 9, 9, y.shape[2]).permute(0, 3, 1, 2).contiguous()


def rnd(x, dt):
    return x.to(dt).float()


def close(got, ref, dt, scale=None, k=1.0):
    scale = float(ref.abs().max()) if scale is None else scale
    tol = (2e-5 if dt == torch.float32 else 1.2e-2) * k * max(scale, 1e-6)
    err = float((got - ref).abs().max())
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e} (scale {scale:.3e})"


def pack(w, dt, mode, n_out, k_in):
    cpk = 32 if dt == torch.bfloat16 else 16
    nbytes = 9 * (k_in // cpk) * (n_out // 16) * 64 * 16
    buf = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    wd = w.contiguous().to(DEV)
    _lib.call("ka_pack_conv3x3", wd, buf, w.shape[0], w.shape[1], n_out, k_in, mode, _lib.dtype_code(dt), st())
    return buf


def run_conv(x_nhwc, wp, B, cin, cout, dt, scale=None, shift=None, bias=None, relu=0, stats=True):
    out = torch.empty(B, 81, cout, dtype=dt, device=DEV)
    rows = _lib.query("ka_conv3x3_sqpart_rows", B)
    bsum = torch.full((B, cout), float("nan"), device=DEV) if stats else None
    sq = torch.full((rows, cout), float("nan"), device=DEV) if stats else None
    _lib.call("ka_conv3x3_fwd", x_nhwc, wp, out, scale, shift, bias, relu, bsum, sq, B, cin, cout,
              _lib.dtype_code(dt), st())
    torch.cuda.synchronize()
    return out, bsum, sq


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("B,cin,cout", [(3, 32, 32), (4, 64, 128), (2, 128, 128), (5, 256, 256), (2, 128, 256)])
def test_conv3x3_forward(dtn, B, cin, cout):
    dt = DT[dtn]
    g = torch.Generator().manual_seed(B * 1000 + cin + cout)
    x = torch.randn(B, cin, 9, 9, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    ref = F.conv2d(rnd(x, dt), rnd(w, dt), padding=1)
    out, bsum, sq = run_conv(to_nhwc(x, dt), pack(w, dt, 0, cout, cin), B, cin, cout, dt)
    close(from_nhwc(out), ref, dt)
    close(bsum.cpu(), ref.sum(dim=(2, 3)), torch.float32, k=20 if dt == torch.float32 else 200)
    close(sq.sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3)), torch.float32, k=20 if dt == torch.float32 else 200)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_conv3x3_fused_input_transform(dtn):
    dt = DT[dtn]
    B, C = 3, 128
    g = torch.Generator().manual_seed(7)
    y = torch.randn(B, C, 9, 9, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    sc, sh = torch.rand(C, generator=g) + 0.5, 0.3 * torch.randn(C, generator=g)
    gb = 0.5 * torch.randn(B, C, generator=g)
    h = torch.relu(rnd(y, dt) * sc[None, :, None, None] + sh[None, :, None, None]) + gb[:, :, None, None]
    ref = F.conv2d(rnd(h, dt), rnd(w, dt), padding=1)
    out, _, _ = run_conv(to_nhwc(y, dt), pack(w, dt, 0, C, C), B, C, C, dt, sc.to(DEV), sh.to(DEV), gb.to(DEV), 1, False)
    close(from_nhwc(out), ref, dt, k=2)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_conv3x3_stem_padded_channels(dtn):
    dt = DT[dtn]
    B, cobs, cpad, C = 3, 50, 64, 64
    g = torch.Generator().manual_seed(9)
    obs = torch.randn(B, cobs, 9, 9, generator=g)
    w = torch.randn(C, cobs, 3, 3, generator=g) / 21.0
    xin = torch.empty(B, 81, cpad, dtype=dt, device=DEV)
    idx = torch.tensor([2, 0, 1], device=DEV)
    _lib.call("ka_obs_to_nhwc", obs.to(DEV), idx, xin, B, cobs, cpad, _lib.dtype_code(dt), st())
    out, _, _ = run_conv(xin, pack(w, dt, 0, C, cpad), B, cpad, C, dt, stats=False)
    ref = F.conv2d(rnd(obs[[2, 0, 1]], dt), rnd(w, dt), padding=1)
    close(from_nhwc(out), ref, dt)
    back = torch.empty(B, C, 9, 9, device=DEV)
    _lib.call("ka_nhwc_to_nchw", out, back, B, C, _lib.dtype_code(dt), st())
    assert torch.equal(back.cpu(), from_nhwc(out))


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("B,cin,cout", [(3, 32, 32), (2, 128, 128), (3, 256, 256)])
def test_conv3x3_dgrad(dtn, B, cin, cout):
    dt = DT[dtn]
    g = torch.Generator().manual_seed(11 + cin)
    dy = torch.randn(B, cout, 9, 9, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cout ** 0.5)
    ref = torch.nn.grad.conv2d_input((B, cin, 9, 9), rnd(w, dt), rnd(dy, dt), padding=1)
    out, bsum, _ = run_conv(to_nhwc(dy, dt), pack(w, dt, 1, cin, cout), B, cout, cin, dt)
    close(from_nhwc(out), ref, dt)
    close(bsum.cpu(), ref.sum(dim=(2, 3)), torch.float32, k=20 if dt == torch.float32 else 200)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("B,cin,cin_real,cout", [(3, 32, 32, 32), (5, 64, 50, 128), (70, 128, 128, 128), (9, 256, 256, 256)])
def test_conv3x3_wgrad(dtn, B, cin, cin_real, cout):
    dt = DT[dtn]
    g = torch.Generator().manual_seed(13 + cin)
    x = torch.randn(B, cin, 9, 9, generator=g)
    x[:, cin_real:] = 0
    dy = torch.randn(B, cout, 9, 9, generator=g) / 9.0
    ref = torch.nn.grad.conv2d_weight(rnd(x, dt), (cout, cin, 3, 3), rnd(dy, dt), padding=1)[:, :cin_real]
    ns = _lib.query("ka_wgrad_splits", B, cin, cout, 0)
    slab = torch.empty(ns * 9 * cout * cin, device=DEV)
    dw = torch.full((cout, cin_real, 3, 3), float("nan"), device=DEV)
    _lib.call("ka_conv3x3_wgrad", to_nhwc(dy, dt), to_nhwc(x, dt), None, None, None, 0, slab, dw, B, cin, cin_real, cout,
              0, 0, _lib.dtype_code(dt), st())
    torch.cuda.synchronize()
    tol_dt = torch.float32 if dt == torch.float32 else torch.bfloat16
    scale = float(ref.abs().max())
    err = float((dw.cpu() - ref).abs().max())
    assert err <= (3e-5 if tol_dt == torch.float32 else 2e-3) * scale, (err, scale)
    # accumulate flag
    _lib.call("ka_conv3x3_wgrad", to_nhwc(dy, dt), to_nhwc(x, dt), None, None, None, 0, slab, dw, B, cin, cin_real, cout,
              1, 0, _lib.dtype_code(dt), st())
    assert float((dw.cpu() - 2 * ref).abs().max()) <= (6e-5 if tol_dt == torch.float32 else 4e-3) * scale


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_wgrad_fused_input_transform(dtn):
    dt = DT[dtn]
    B, C = 4, 128
    g = torch.Generator().manual_seed(17)
    y = torch.randn(B, C, 9, 9, generator=g)
    dy = torch.randn(B, C, 9, 9, generator=g) / 9
    sc, sh = torch.rand(C, generator=g) + 0.5, 0.3 * torch.randn(C, generator=g)
    gb = 0.5 * torch.randn(B, C, generator=g)
    h = rnd(torch.relu(rnd(y, dt) * sc[None, :, None, None] + sh[None, :, None, None]) + gb[:, :, None, None], dt)
    ref = torch.nn.grad.conv2d_weight(h, (C, C, 3, 3), rnd(dy, dt), padding=1)
    ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
    slab = torch.empty(ns * 9 * C * C, device=DEV)
    dw = torch.empty(C, C, 3, 3, device=DEV)
    _lib.call("ka_conv3x3_wgrad", to_nhwc(dy, dt), to_nhwc(y, dt), sc.to(DEV), sh.to(DEV), gb.to(DEV), 1, slab, dw, B, C, C, C,
              0, 0, _lib.dtype_code(dt), st())
    scale = float(ref.abs().max())
    assert float((dw.cpu() - ref).abs().max()) <= (3e-5 if dt == torch.float32 else 3e-3) * scale


# ------------------------------------------------------------------ BN + tail kernels

@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("C", [32, 128, 256])
def test_bn_stats_and_tail_forward(ka_env, dtn, C):
    dt = DT[dtn]
    B = 5
    g = torch.Generator().manual_seed(23 + C)
    y = torch.randn(B, C, 9, 9, generator=g) * 1.7 + 0.3
    x = torch.relu(torch.randn(B, C, 9, 9, generator=g))
    x[1, 3] = 0.0
    se = torch.randn(B, 2 * C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    yq, xq = rnd(y, dt), rnd(x, dt)
    # statistics from per-board sums (as the conv epilogue would produce them)
    bsum = yq.sum(dim=(2, 3)).to(DEV)
    sq = (yq ** 2).sum(dim=(0, 2, 3))[None].to(DEV)
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    ws = torch.empty(_lib.query("ka_reduce_workspace_doubles", C), dtype=torch.float64, device=DEV)
    _lib.call("ka_bn_reduce", bsum, B, sq, 1, C, sums, ws, st())
    scale, shift, mean, invstd = (torch.empty(C, device=DEV) for _ in range(4))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    _lib.call("ka_bn_coeffs", sums, float(B * 81), None, gamma.to(DEV), beta.to(DEV), rmd, rvd, nbt, 0.1, 1e-5, scale, shift,
              mean, invstd, C, st())
    assert int(nbt) == 1
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(yq, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    zk = yq * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]
    assert torch.allclose(zk, z, rtol=1e-4, atol=1e-5)
    # tail
    ref = torch.relu(z * torch.sigmoid(se[:, :C])[:, :, None, None] + se[:, C:, None, None] + xq)
    out = torch.empty(B, 81, C, dtype=dt, device=DEV)
    pool = torch.empty(B, 4 * C, device=DEV)
    _lib.call("ka_block_tail_fwd", to_nhwc(y, dt), scale, shift, se.to(DEV), to_nhwc(x, dt), out, pool, B, C,
              _lib.dtype_code(dt), st())
    got = from_nhwc(out)
    close(got, ref, dt, k=2)
    assert torch.allclose(pool.cpu()[:, :3 * C], orc.global_pool(got), rtol=1e-5, atol=1e-5)
    ties = (got == got.amax(dim=(2, 3), keepdim=True)).sum(dim=(2, 3)).float()
    assert torch.equal(pool.cpu()[:, 3 * C:], ties)
    pool2 = torch.empty(B, 4 * C, device=DEV)
    _lib.call("ka_pool_fwd", out, pool2, B, C, _lib.dtype_code(dt), st())
    assert torch.equal(pool2, pool)
    # the forms that request the loads of 2 / 3 / all 6 of a thread's squares together (KA_TAIL_FWD_KB): the same bits
    for kb in ("0", "2", "3", "6"):
        ka_env.set("KA_TAIL_FWD_KB", kb)
        out_k = torch.full_like(out, float("nan")); pool_k = torch.full_like(pool, float("nan"))
        _lib.call("ka_block_tail_fwd", to_nhwc(y, dt), scale, shift, se.to(DEV), to_nhwc(x, dt), out_k, pool_k, B, C,
                  _lib.dtype_code(dt), st())
        torch.cuda.synchronize()
        assert torch.equal(out_k, out) and torch.equal(pool_k, pool), kb
    ka_env.unset("KA_TAIL_FWD_KB")
    # the squeeze-excite chain inside the launch (ka_block_tail_fwd_se): se from the per-board sums through the two FC layers,
    # then the same tail -- against the chain in float64 torch + ka_block_tail_fwd on the same operands
    H = max(C // 16, 4)
    code = _lib.dtype_code(dt)
    if _lib.query("ka_block_tail_fwd_se_supported", C, H, code):
        W1, b1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(DEV), torch.randn(H, generator=g).to(DEV)
        W2, b2 = (torch.randn(2 * C, H, generator=g) / H ** 0.5).to(DEV), torch.randn(2 * C, generator=g).to(DEV)
        sqz_r = scale * (bsum * (1.0 / 81.0)) + shift
        se1_r = torch.relu(sqz_r.double() @ W1.double().t() + b1.double()).float()
        se_r = (se1_r.double() @ W2.double().t() + b2.double()).float().contiguous()
        out_r = torch.empty_like(out); pool_r = torch.empty_like(pool)
        _lib.call("ka_block_tail_fwd", to_nhwc(y, dt), scale, shift, se_r, to_nhwc(x, dt), out_r, pool_r, B, C, code, st())
        sqz, se1, se2 = torch.empty(B, C, device=DEV), torch.empty(B, H, device=DEV), torch.empty(B, 2 * C, device=DEV)
        out_s = torch.full_like(out, float("nan")); pool_s = torch.full_like(pool, float("nan"))
        _lib.call("ka_block_tail_fwd_se", to_nhwc(y, dt), scale, shift, bsum, W1, b1, W2, b2, to_nhwc(x, dt), out_s, pool_s,
                  sqz, se1, se2, B, C, H, code, st())
        torch.cuda.synchronize()
        assert torch.allclose(sqz, sqz_r, rtol=1e-6, atol=1e-6)
        assert torch.allclose(se1, se1_r, rtol=1e-5, atol=2e-6) and torch.allclose(se2, se_r, rtol=1e-5, atol=5e-6)
        # the tail itself on the launch's OWN se is the plain launch bit for bit
        out_t = torch.empty_like(out); pool_t = torch.empty_like(pool)
        _lib.call("ka_block_tail_fwd", to_nhwc(y, dt), scale, shift, se2, to_nhwc(x, dt), out_t, pool_t, B, C, code, st())
        torch.cuda.synchronize()
        assert torch.equal(out_s, out_t) and torch.equal(pool_s, pool_t)
        close(from_nhwc(out_s), from_nhwc(out_r), dt, k=2)
    else:
        assert C == 32                                            # (2 C <= threads of the workgroup fails only there)
    # eval coefficients
    es, eh = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.call("ka_bn_eval_coeffs", gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5, es, eh, C, st())
    ze = F.batch_norm(yq, rm, rv, gamma, beta, training=False, eps=1e-5)
    assert torch.allclose(yq * es.cpu()[None, :, None, None] + eh.cpu()[None, :, None, None], ze, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("C", [32, 256])
def test_block_tail_backward_chain(dtn, C):
    """tail_bwd_reduce -> tail_bwd_dz -> bn_bwd_coeffs -> bn_bwd_apply  ==  autograd through
    relu(bn(y)*sigmoid(a)+b+x) w.r.t. y, a, b, gamma, beta (SE squeeze path via dsq)."""
    dt = DT[dtn]
    B = 4
    g = torch.Generator().manual_seed(31 + C)
    y = rnd(torch.randn(B, C, 9, 9, generator=g), dt).requires_grad_(True)
    x = rnd(torch.relu(torch.randn(B, C, 9, 9, generator=g)), dt)
    se = torch.randn(B, 2 * C, generator=g).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    dsq = 0.1 * torch.randn(B, C, generator=g)        # gradient arriving at mean_p(z) from the SE FCs
    dout = rnd(torch.randn(B, C, 9, 9, generator=g), dt)
    z = F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5)
    out = torch.relu(z * torch.sigmoid(se[:, :C])[:, :, None, None] + se[:, C:, None, None] + x)
    loss = (out * dout).sum() + (z.mean(dim=(2, 3)) * dsq).sum()
    gy, gse, gg, gb = torch.autograd.grad(loss, [y, se, gamma, beta])
    # device side
    yd, outd, doutd = to_nhwc(y.detach(), dt), to_nhwc(rnd(out.detach(), dt), dt), to_nhwc(dout, dt)
    mu = y.detach().mean(dim=(0, 2, 3)); var = y.detach().var(dim=(0, 2, 3), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    scale = (gamma.detach() * invstd).to(DEV); shift = (beta.detach() - mu * gamma.detach() * invstd).to(DEV)
    dse = torch.empty(B, 2 * C, device=DEV)
    _lib.call("ka_tail_bwd_reduce", doutd, outd, yd, scale, shift, se.detach().to(DEV), dse, B, C, _lib.dtype_code(dt), st())
    close(dse.cpu(), gse, dt, k=3)
    dz = torch.empty(B, 81, C, dtype=dt, device=DEV)
    s1, s2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    _lib.call("ka_tail_bwd_dz", doutd, outd, yd, se.detach().to(DEV), dsq.to(DEV), mu.to(DEV), invstd.to(DEV), dz, s1, s2,
              B, C, _lib.dtype_code(dt), st())
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    ws = torch.empty(_lib.query("ka_reduce_workspace_doubles", C), dtype=torch.float64, device=DEV)
    _lib.call("ka_pair_reduce", s1, s2, B, C, sums, ws, st())
    dgam, dbet, k = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(3 * C, device=DEV)
    _lib.call("ka_bn_bwd_coeffs", sums, sums, float(B * 81), None, gamma.detach().to(DEV), mu.to(DEV), invstd.to(DEV), dgam,
              dbet, k, C, 1, st())
    close(dgam.cpu(), gg, dt, k=3); close(dbet.cpu(), gb, dt, k=3)
    dy = torch.empty(B, 81, C, dtype=dt, device=DEV)
    _lib.call("ka_bn_bwd_apply", dz, yd, k, dy, B, C, _lib.dtype_code(dt), st())
    close(from_nhwc(dy), gy, dt, k=3)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_relu_bn_backward_and_block_dx(dtn):
    dt = DT[dtn]
    B, C = 4, 128
    g = torch.Generator().manual_seed(41)
    y = rnd(torch.randn(B, C, 9, 9, generator=g), dt).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    dh = rnd(torch.randn(B, C, 9, 9, generator=g), dt)
    h = torch.relu(F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5))
    gy, gg, gb = torch.autograd.grad((h * dh).sum(), [y, gamma, beta])
    mu = y.detach().mean(dim=(0, 2, 3)); invstd = 1 / torch.sqrt(y.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    scale = (gamma.detach() * invstd).to(DEV); shift = (beta.detach() - mu * gamma.detach() * invstd).to(DEV)
    yd = to_nhwc(y.detach(), dt)
    da = torch.empty(B, 81, C, dtype=dt, device=DEV)
    s1, s2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    _lib.call("ka_relu_bn_bwd_reduce", to_nhwc(dh, dt), yd, scale, shift, mu.to(DEV), invstd.to(DEV), da, s1, s2, B, C,
              _lib.dtype_code(dt), st())
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    ws = torch.empty(_lib.query("ka_reduce_workspace_doubles", C), dtype=torch.float64, device=DEV)
    _lib.call("ka_pair_reduce", s1, s2, B, C, sums, ws, st())
    dgam, dbet, k = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(3 * C, device=DEV)
    _lib.call("ka_bn_bwd_coeffs", sums, sums, float(B * 81), None, gamma.detach().to(DEV), mu.to(DEV), invstd.to(DEV), dgam,
              dbet, k, C, 1, st())
    dy = torch.empty(B, 81, C, dtype=dt, device=DEV)
    _lib.call("ka_bn_bwd_apply", da, yd, k, dy, B, C, _lib.dtype_code(dt), st())
    close(from_nhwc(dy), gy, dt, k=3); close(dgam.cpu(), gg, dt, k=3); close(dbet.cpu(), gb, dt, k=3)

    # block_dx: residual + pool backward with ties / dead channels / constant planes
    x = rnd(torch.relu(torch.randn(B, C, 9, 9, generator=g)), dt)
    x[1, 3] = 0.0; x[2, 5] = 0.75; x[0, 7, 0, :3] = 9.0
    xr = x.clone().requires_grad_(True)
    dpool = torch.randn(B, 3 * C, generator=g)
    dxc = rnd(torch.randn(B, C, 9, 9, generator=g), dt)
    dout = rnd(torch.randn(B, C, 9, 9, generator=g), dt)
    outv = rnd(torch.randn(B, C, 9, 9, generator=g), dt)
    ref = torch.autograd.grad((orc.global_pool(xr) * dpool).sum(), xr)[0] + dxc + dout * (outv > 0)
    dx = torch.empty(B, 81, C, dtype=dt, device=DEV)
    xpool = torch.empty(B, 4 * C, device=DEV)
    _lib.call("ka_pool_fwd", to_nhwc(x, dt), xpool, B, C, _lib.dtype_code(dt), st())
    _lib.call("ka_block_dx", to_nhwc(dxc, dt), to_nhwc(dout, dt), to_nhwc(outv, dt), to_nhwc(x, dt), xpool, dpool.to(DEV), dx,
              B, C, _lib.dtype_code(dt), st())
    close(from_nhwc(dx), ref, dt, k=2)
    dx2 = torch.empty(B, 81, C, dtype=dt, device=DEV)
    _lib.call("ka_block_dx", None, None, None, to_nhwc(x, dt), xpool, dpool.to(DEV), dx2, B, C, _lib.dtype_code(dt), st())
    close(from_nhwc(dx2), ref - dxc - dout * (outv > 0), dt, k=2)


# ------------------------------------------------------------------ small GEMMs

@pytest.mark.parametrize("M,N,K,ta,tb", [(37, 50, 70, 0, 1), (130, 16, 768, 0, 1), (65, 96, 33, 0, 0), (40, 24, 1000, 1, 0),
                                        (810, 32, 139, 0, 0), (139, 32, 1701, 1, 0), (243, 139, 32, 0, 1), (70, 33, 45, 1, 1)])
def test_gemm_variants(M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.t() if ta else A) @ (Bm.t() if tb else Bm)
    C = torch.empty(M, N, device=DEV)
    _lib.call("ka_gemm", A.to(DEV), Bm.to(DEV), C, bias.to(DEV), M, N, K, A.shape[1], Bm.shape[1], N, ta, tb, 0, 0, 0, 1, 0, 1, st())
    assert torch.allclose(C.cpu(), torch.relu(ref + bias), rtol=1e-5, atol=1e-4)
    # split-K + reduce, bf16 operand
    ns = 3
    slab = torch.empty(ns, M, N, device=DEV)
    _lib.call("ka_gemm", A.bfloat16().to(DEV), Bm.to(DEV), slab, None, M, N, K, A.shape[1], Bm.shape[1], N, ta, tb, 1, 0, 0, 0, 0, ns, st())
    out = torch.ones(M, N, device=DEV)
    _lib.call("ka_reduce_slabs", slab, out, ns, M * N, 1, st())
    ref2 = (A.bfloat16().float().t() if ta else A.bfloat16().float()) @ (Bm.t() if tb else Bm)
    assert torch.allclose(out.cpu(), ref2 + 1, rtol=1e-4, atol=1e-3)
    # bf16 output
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_gemm", A.to(DEV), Bm.to(DEV), Cb, None, M, N, K, A.shape[1], Bm.shape[1], N, ta, tb, 0, 0, 1, 0, 0, 1, st())
    assert torch.allclose(Cb.float().cpu(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    # bf16 B operand, accumulate into C
    Cacc = torch.full((M, N), 2.0, device=DEV)
    _lib.call("ka_gemm", A.to(DEV), Bm.bfloat16().to(DEV), Cacc, None, M, N, K, A.shape[1], Bm.shape[1], N, ta, tb, 0, 1, 0, 0, 1, 1, st())
    ref3 = (A.t() if ta else A) @ (Bm.bfloat16().float().t() if tb else Bm.bfloat16().float())
    assert torch.allclose(Cacc.cpu(), ref3 + 2, rtol=1e-4, atol=1e-3)


def test_row_kernels():
    g = torch.Generator().manual_seed(5)
    M, N = 1000, 24
    A, Bm = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    p1, p2 = torch.empty(4, N, device=DEV), torch.empty(4, N, device=DEV)
    _lib.call("ka_colsum", A.to(DEV), Bm.to(DEV), p1, p2, M, N, 4, st())
    assert torch.allclose(p1.sum(0).cpu(), A.sum(0), atol=1e-3) and torch.allclose(p2.sum(0).cpu(), (A * Bm).sum(0), atol=1e-3)
    _lib.call("ka_rows_sq_sums", A.to(DEV), p1, p2, M, N, 4, st())
    assert torch.allclose(p2.sum(0).cpu(), (A * A).sum(0), atol=1e-3)
    gd, h = A.clone().to(DEV), Bm.to(DEV)
    _lib.call("ka_relu_mask", gd, h, M * N, st())
    assert torch.equal(gd.cpu(), A * (Bm > 0))
    sc, sh = torch.rand(N, generator=g), torch.randn(N, generator=g)
    out = torch.empty(M, N, device=DEV)
    _lib.call("ka_rows_affine_relu", A.to(DEV), sc.to(DEV), sh.to(DEV), out, M, N, st())
    assert torch.allclose(out.cpu(), torch.relu(A * sc + sh), atol=1e-6)
    af = torch.empty(7, N, device=DEV)
    _lib.call("ka_affine_rows", A[:7].contiguous().to(DEV), sc.to(DEV), sh.to(DEV), 0.5, af, 7, N, st())
    assert torch.allclose(af.cpu(), sc * (A[:7] * 0.5) + sh, atol=1e-6)


# ------------------------------------------------------------------ GAE / loss / optimiser vs golden

def test_gae_bit_exact(golden):
    g = golden("g4_gae")
    r, v, nv = g["rewards"].to(DEV), g["values"].to(DEV), g["next_value"].to(DEV)
    T, N = r.shape
    adv = torch.empty_like(r)

    def run(term, ov=None, ln=None):
        _lib.call("ka_gae", r, v, term.float().to(DEV), nv, None if ov is None else ov.to(DEV),
                  None if ln is None else ln.to(DEV), adv, T, N, 0.99, 0.95, 0, st())
        return adv.cpu().numpy()

    assert np.array_equal(run(g["terminated"]), g.np("adv_gpu"))
    assert np.array_equal(run(g["terminated"], g["override"]), g.np("adv_override_gpu"))
    assert np.array_equal(run(g["terminated_padded"], None, g["lengths"]), g.np("adv_padded_gpu"))
    assert np.allclose(run(g["terminated_padded"], g["override"], g["lengths"]), g.np("adv_padded_override"), rtol=1e-5, atol=1e-5)
    # f64
    r64, v64 = r[:16, :4].double().contiguous(), v[:16, :4].double().contiguous()
    a64 = torch.empty_like(r64)
    _lib.call("ka_gae", r64, v64, g["terminated"][:16, :4].float().contiguous().to(DEV), nv[:4].double().contiguous(), None,
              None, a64, 16, 4, 0.99, 0.95, 1, st())
    assert np.allclose(a64.cpu().numpy(), g.np("adv_f64"), rtol=1e-12, atol=1e-12)
    # advantage normalisation (unbiased std)
    flat = torch.from_numpy(g.np("adv_gpu")).reshape(-1)
    out = torch.empty_like(flat, device=DEV)
    _lib.call("ka_normalize_advantages", flat.to(DEV), out, flat.numel(), st())
    assert torch.allclose(out.cpu(), orc.normalize_advantages(flat), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["third.", "ragged."])
def test_fused_loss_matches_reference(golden, tag):
    g = golden("g3_loss")
    B, A = 4, 11259
    logits = g[tag + "logits"].reshape(B, A).to(DEV)
    dl = torch.empty(B, A, device=DEV)
    nlp, rl, re = (torch.empty(B, device=DEV) for _ in range(3))
    flags = torch.zeros(2, dtype=torch.int32, device=DEV)
    lp, lv, ls, ce, eps = 1.0, 1.5, 0.1, 0.01, 0.2
    _lib.call("ka_policy_loss", logits, g[tag + "legal"].to(DEV), g[tag + "actions"].to(DEV), g[tag + "old_log_probs"].to(DEV),
              g[tag + "advantages"].to(DEV), None, dl, nlp, rl, re, flags, None, eps, lp / B, ce / B, B, A, 0, st())
    out = torch.zeros(16, device=DEV)
    acc = torch.zeros(4, device=DEV)
    dv, ds = torch.empty(B, 3, device=DEV), torch.empty(B, device=DEV)
    _lib.call("ka_value_loss", g[tag + "value_logits"].to(DEV), g[tag + "score"].reshape(B).to(DEV), g[tag + "value_cats"].to(DEV),
              g[tag + "score_targets"].to(DEV), None, rl, re, dv, ds, out, acc, None, lp, lv, ls, ce, 1, B, st())
    o = out.cpu()
    assert flags.cpu().tolist() == [0, 0]
    assert torch.allclose(nlp.cpu(), g[tag + "new_log_probs"], rtol=1e-5, atol=1e-5)
    for i, k in enumerate(["policy_loss", "value_loss", "score_loss", "entropy", "total"]):
        assert abs(float(o[i]) - float(g[tag + k])) <= 1e-5 * max(1.0, abs(float(g[tag + k]))), k
    assert torch.allclose(dl.cpu(), g[tag + "grad.logits"].reshape(B, A), rtol=1e-4, atol=1e-8)
    assert torch.allclose(dv.cpu(), g[tag + "grad.value_logits"], rtol=1e-4, atol=1e-8)
    assert torch.allclose(ds.cpu(), g[tag + "grad.score"].reshape(B), rtol=1e-4, atol=1e-8)
    assert abs(float(acc[1]) - (lv * float(g[tag + "value_loss"]) + ls * float(g[tag + "score_loss"]))) < 1e-5
    # scalar value projection
    sv = torch.empty(B, device=DEV)
    _lib.call("ka_scalar_value", g[tag + "value_logits"].to(DEV), (g[tag + "score"] * 3).reshape(B).to(DEV), 0.1, sv, B, st())
    assert torch.allclose(sv.cpu(), g[tag + "scalar_blended"], rtol=1e-5, atol=1e-6)


def test_loss_guards_and_gather():
    B, A, S = 3, 11259, 6
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B, A, generator=g)
    mb = orc.synth_minibatch(S, seed=9, legal_kind="ragged")
    idx = torch.tensor([4, 1, 5])
    old = mb["old_log_probs"]
    ref_lp, ref_ent = orc.masked_policy_terms(logits, mb["legal"][idx], mb["actions"][idx])
    nlp, rl, re = (torch.empty(B, device=DEV) for _ in range(3))
    flags = torch.zeros(2, dtype=torch.int32, device=DEV)
    args = lambda lg, legal: ("ka_policy_loss", lg.to(DEV), legal.to(DEV), mb["actions"].to(DEV), old.to(DEV),
                              mb["advantages"].to(DEV), idx.to(DEV), None, nlp, rl, re, flags, None, 0.2, 1.0 / B, 0.01 / B, B, A, 0, st())
    _lib.call(*args(logits, mb["legal"]))
    assert flags.cpu().tolist() == [0, 0]
    assert torch.allclose(nlp.cpu(), ref_lp, rtol=1e-5, atol=1e-5)
    assert abs(float(re.mean()) - float(ref_ent)) < 1e-5
    bad = logits.clone(); bad[1, 77] = float("nan")
    _lib.call(*args(bad, mb["legal"]))
    assert flags.cpu().tolist() == [1, 0]
    flags.zero_()
    legal = mb["legal"].clone(); legal[5] = False
    _lib.call(*args(logits, legal))
    assert flags.cpu().tolist() == [0, 1]


def test_clip_adam_matches_reference(golden):
    g = golden("g6_adam")
    n = 5
    params = [g[f"p0.{i}"].clone().to(DEV) for i in range(n)]
    grads = [torch.empty_like(p) for p in params]
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    chunk = _lib.query("ka_adam_chunk")
    recs, bt, bo = [], [], []
    for i, p in enumerate(params):
        recs += [p.data_ptr(), grads[i].data_ptr(), ms[i].data_ptr(), vs[i].data_ptr(), p.numel()]
        for off in range(0, p.numel(), chunk):
            bt.append(i); bo.append(off)
    tab = torch.tensor(recs, dtype=torch.int64, device=DEV)
    btd, bod = torch.tensor(bt, dtype=torch.int32, device=DEV), torch.tensor(bo, dtype=torch.int64, device=DEV)
    partial = torch.empty(len(bt), dtype=torch.float64, device=DEV)
    ctl, step = torch.zeros(4, device=DEV), torch.zeros(1, device=DEV)
    accn = torch.zeros(1, device=DEV)
    for s in range(3):
        for i in range(n):
            grads[i].copy_(g[f"g{s}.{i}"])
        _lib.call("ka_clip_adam_step", tab, btd, bod, len(bt), partial, ctl, step, None, None, accn, 1.0, 2e-4, 0.9, 0.999, 1e-8, st())
        assert abs(float(ctl[0]) - float(g[f"norm{s}"])) <= 1e-5 * float(g[f"norm{s}"])
        for i in range(n):
            assert torch.allclose(params[i].cpu(), g[f"p{s + 1}.{i}"], rtol=1e-6, atol=1e-7), (s, i)
    assert float(step) == 3.0
    for i in range(n):
        assert torch.allclose(ms[i].cpu(), g[f"m.{i}"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(vs[i].cpu(), g[f"v.{i}"], rtol=1e-5, atol=1e-10)
    # inf gradient -> step skipped, scaler backs off; guard flag -> skipped too
    before = [p.clone() for p in params]
    grads[0][0] = float("inf")
    scaler = torch.tensor([65536.0, 5.0], device=DEV)
    _lib.call("ka_clip_adam_step", tab, btd, bod, len(bt), partial, ctl, step, scaler, None, None, 1.0, 2e-4, 0.9, 0.999, 1e-8, st())
    assert float(step) == 3.0 and all(torch.equal(a, b) for a, b in zip(before, params))
    assert scaler.cpu().tolist() == [32768.0, 0.0]
    grads[0][0] = 1.0
    flags = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    _lib.call("ka_clip_adam_step", tab, btd, bod, len(bt), partial, ctl, step, scaler, flags, None, 1.0, 2e-4, 0.9, 0.999, 1e-8, st())
    assert float(step) == 3.0 and all(torch.equal(a, b) for a, b in zip(before, params))
    assert scaler.cpu().tolist() == [32768.0, 1.0]


def test_conv3x3_dgrad_fused_matches_unfused_sequence():
    """ka_conv3x3_dgrad_fused == bn_bwd_apply -> conv (dgrad pack) -> relu_bn_bwd_reduce, incl. the dy side output."""
    dt, code = torch.bfloat16, 1
    B, C = 5, 128
    g = torch.Generator().manual_seed(71)
    dz = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    y = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    yprev = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    k = torch.cat([torch.rand(C, generator=g) + 0.5, 0.1 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)]).to(DEV)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    wp = pack(w, dt, 1, C, C)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    mu, istd = (0.1 * torch.randn(C, generator=g)).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    rows = _lib.query("ka_conv3x3_sqpart_rows", B)
    # unfused reference sequence
    dy_ref = torch.empty_like(dz)
    _lib.call("ka_bn_bwd_apply", dz, y, k, dy_ref, B, C, code, st())
    dh_ref, dg_ref, _ = run_conv(dy_ref, wp, B, C, C, dt)
    da_ref = torch.empty_like(dz); s1, s2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    _lib.call("ka_relu_bn_bwd_reduce", dh_ref, yprev, sc, sh, mu, istd, da_ref, s1, s2, B, C, code, st())
    # fused
    dy, da, dg = torch.empty_like(dz), torch.empty_like(dz), torch.empty(B, C, device=DEV)
    e1, e2 = torch.empty(rows, C, device=DEV), torch.empty(rows, C, device=DEV)
    _lib.call("ka_conv3x3_dgrad_fused", dz, y, k, dy, wp, da, dg, yprev, sc, sh, mu, istd, e1, e2, B, C, C, code, st())
    torch.cuda.synchronize()
    assert torch.equal(dy, dy_ref)
    assert torch.allclose(dg, dg_ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(da, da_ref)
    # partial sums are taken from the bf16-rounded dh in both paths
    assert torch.allclose(e1.sum(0), s1.sum(0), rtol=1e-4, atol=1e-3) and torch.allclose(e2.sum(0), s2.sum(0), rtol=1e-4, atol=1e-3)
    # without the epilogue and without dy side output
    out2 = torch.empty_like(dz)
    _lib.call("ka_conv3x3_dgrad_fused", dz, y, k, None, wp, out2, None, None, None, None, None, None, None, None, B, C, C, code, st())
    assert torch.equal(out2, dh_ref)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("B,C,H", [(3, 32, 4), (5, 128, 8), (4, 256, 16)])
def test_tail_bwd_fused_matches_unfused_sequence(dtn, B, C, H):
    """ka_tail_bwd_fused == tail_bwd_reduce -> se_fc2 backward -> ReLU mask -> se_fc1 backward -> tail_bwd_dz."""
    dt = DT[dtn]
    code = _lib.dtype_code(dt)
    assert _lib.query("ka_tail_bwd_fused_supported", C, H, code)
    g = torch.Generator().manual_seed(900 + C)
    dout = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    out = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    y = to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    mu, istd = (0.1 * torch.randn(C, generator=g)).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    se = torch.randn(B, 2 * C, generator=g).to(DEV)
    se1 = torch.randn(B, H, generator=g).to(DEV)
    W2 = (torch.randn(2 * C, H, generator=g) / H ** 0.5).to(DEV)
    W1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(DEV)
    # unfused reference sequence (the FC steps in fp64 torch)
    dse_ref = torch.empty(B, 2 * C, device=DEV)
    _lib.call("ka_tail_bwd_reduce", dout, out, y, sc, sh, se, dse_ref, B, C, code, st())
    dh_ref = ((dse_ref.double() @ W2.double()) * (se1 > 0)).float()
    dsq_ref = (dh_ref.double() @ W1.double()).float().contiguous()
    dz_ref = torch.empty_like(dout); s1r, s2r = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    _lib.call("ka_tail_bwd_dz", dout, out, y, se, dsq_ref, mu, istd, dz_ref, s1r, s2r, B, C, code, st())
    # fused
    dz = torch.empty_like(dout); dse = torch.empty(B, 2 * C, device=DEV); dh = torch.empty(B, H, device=DEV)
    s1, s2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    _lib.call("ka_tail_bwd_fused", dout, out, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2, B, C, H, code, st())
    torch.cuda.synchronize()
    close(dse.cpu(), dse_ref.cpu(), torch.float32, k=5)
    close(dh.cpu(), dh_ref.cpu(), torch.float32, k=5)
    close(dz.float().cpu(), dz_ref.float().cpu(), dt, k=1 if dt == torch.float32 else 0.5)
    close(s1.cpu(), s1r.cpu(), torch.float32, k=10)
    close(s2.cpu(), s2r.cpu(), torch.float32, k=10)
    assert not _lib.query("ka_tail_bwd_fused_supported", 48, 5, code)


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
@pytest.mark.parametrize("B,C,H,heads", [(3, 32, 4, False), (5, 128, 8, True), (4, 256, 16, False), (515, 256, 16, False)])
def test_block_dx_tail_bwd_equals_the_two_launches(ka_env, dtn, B, C, H, heads):
    """ka_block_dx_tail_bwd == ka_block_dx followed by ka_tail_bwd_fused (dout = its dx, out = its x), every output bit for bit;
    `heads`: the form that enters the tower (no residual branch above)."""
    dt = DT[dtn]
    code = _lib.dtype_code(dt)
    if not _lib.query("ka_block_dx_tail_bwd_supported", C, H, code):
        pytest.skip("shape not covered by the fused boundary launch")
    g = torch.Generator().manual_seed(77 + C + B)
    A = lambda: to_nhwc(torch.randn(B, C, 9, 9, generator=g), dt)
    dxc, dout_up, out_up, y = A(), A(), A(), A()
    x = to_nhwc(torch.relu(torch.randn(B, C, 9, 9, generator=g)), dt)          # a block output: ReLU'd, with exact zeros and ties
    pool = torch.empty(B, 4 * C, device=DEV)
    _lib.call("ka_pool_fwd", x, pool, B, C, code, st())
    dpool = torch.randn(B, 3 * C, generator=g).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    mu, istd = (0.1 * torch.randn(C, generator=g)).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    se, se1 = torch.randn(B, 2 * C, generator=g).to(DEV), torch.randn(B, H, generator=g).to(DEV)
    W2 = (torch.randn(2 * C, H, generator=g) / H ** 0.5).to(DEV)
    W1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(DEV)
    up = (None, None) if heads else (dout_up, out_up)

    def outs():
        return (torch.empty_like(x), torch.empty_like(x), torch.empty(B, 2 * C, device=DEV), torch.empty(B, H, device=DEV),
                torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV))
    dx_r, dz_r, dse_r, dh_r, s1_r, s2_r = outs()
    _lib.call("ka_block_dx", dxc, *up, x, pool, dpool, dx_r, B, C, code, st())
    _lib.call("ka_tail_bwd_fused", dx_r, x, y, sc, sh, se, se1, W2, W1, mu, istd, dz_r, dse_r, dh_r, s1_r, s2_r, B, C, H, code, st())
    dx, dz, dse, dh, s1, s2 = outs()
    _lib.call("ka_block_dx_tail_bwd", dxc, *up, x, pool, dpool, dx, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2,
              B, C, H, code, st())
    torch.cuda.synchronize()
    for name, a, b in (("dx", dx, dx_r), ("dz", dz, dz_r), ("dse", dse, dse_r), ("dh", dh, dh_r), ("s1", s1, s1_r), ("s2", s2, s2_r)):
        assert torch.equal(a, b), name
    # the chain form: takes the masked gradient of the block above (no out_up read), writes du = dx * [x > 0]
    du_up = None if heads else torch.where(out_up > 0, dout_up, torch.zeros_like(dout_up))
    du, dz, dse, dh, s1, s2 = outs()
    _lib.call("ka_block_dx_tail_bwd_du", dxc, du_up, x, pool, dpool, du, y, sc, sh, se, se1, W2, W1, mu, istd, dz, dse, dh, s1, s2,
              B, C, H, code, st())
    torch.cuda.synchronize()
    du_r = torch.where(x > 0, dx_r, torch.zeros_like(dx_r))
    for name, a, b in (("du", du, du_r), ("dz", dz, dz_r), ("dse", dse, dse_r), ("dh", dh, dh_r), ("s1", s1, s1_r), ("s2", s2, s2_r)):
        assert torch.equal(a, b), name
    # the chain form without dz: same du / dse / dh / s1 / s2, and bf16(fmaf(du, gate, add)) IS the dz of the other forms
    ka_env.set("KA_TAIL_GATE_P4", "0")
    du_g, _, dse, dh, s1, s2 = outs()
    gate_add = torch.full((2, B, C), float("nan"), device=DEV)
    _lib.call("ka_block_dx_tail_bwd_du_gate", dxc, du_up, x, pool, dpool, du_g, y, sc, sh, se, se1, W2, W1, mu, istd,
              gate_add[0], gate_add[1], dse, dh, s1, s2, B, C, H, code, st())
    torch.cuda.synchronize()
    for name, a, b in (("du", du_g, du_r), ("dse", dse, dse_r), ("dh", dh, dh_r), ("s1", s1, s1_r), ("s2", s2, s2_r)):
        assert torch.equal(a, b), "gate form: " + name
    assert float((gate_add[0] - torch.sigmoid(se[:, :C])).abs().max()) < 1e-6
    dz_g = torch.addcmul(gate_add[1].double()[:, None, :], du_g.double(), gate_add[0].double()[:, None, :])   # one rounding, like fmaf
    assert torch.equal(dz_g.float().to(dt), dz_r), "gate form: dz"
    # the half-width form of that launch (a thread owns four channels and every eighth square; the default for 512-thread bf16
    # shapes, KA_TAIL_GATE_P4=0 selects the other): the same du, sums in another order
    for p4 in ("0", "1"):
        ka_env.set("KA_TAIL_GATE_P4", p4)
        du_h, _, dse_h, dh_h, s1_h, s2_h = outs()
        ga_h = torch.full((2, B, C), float("nan"), device=DEV)
        _lib.call("ka_block_dx_tail_bwd_du_gate", dxc, du_up, x, pool, dpool, du_h, y, sc, sh, se, se1, W2, W1, mu, istd,
                  ga_h[0], ga_h[1], dse_h, dh_h, s1_h, s2_h, B, C, H, code, st())
        torch.cuda.synchronize()
        assert torch.equal(du_h, du_r), p4
        for name, a, b in (("dse", dse_h, dse_r), ("dh", dh_h, dh_r), ("s1", s1_h, s1_r), ("s2", s2_h, s2_r), ("gate_add", ga_h, gate_add)):
            assert not bool(a.isnan().any()), (p4, name)
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, (p4, name)
    ka_env.unset("KA_TAIL_GATE_P4")
    # ... and ka_block_dx takes a du as its dout: same result as from the unmasked gradient
    if not heads:
        dx2 = torch.empty_like(x)
        _lib.call("ka_block_dx", dxc, du_up, out_up, x, pool, dpool, dx2, B, C, code, st())
        torch.cuda.synchronize()
        assert torch.equal(dx2, dx_r)


@pytest.mark.parametrize("B,C", [(515, 256), (1024, 256), (4096, 256), (515, 128), (2048, 128)])
def test_gated_data_gradient_equals_the_plain_one_on_its_own_dy(B, C):
    """ka_conv3x3_dgrad_fused_gated(du, [gate | add], y, k): the input transform forms dz = du * gate[b, c] + add[b, c] in fp32 and
    dy = dz * k0 + k1 + y * k2 from it.  (1) Its written-back dy is the fp32 formula rounded once to bf16 (the dz form rounds
    twice); (2) everything downstream of the transform is the same kernel: ka_conv3x3_dgrad_fused fed that dy with the identity
    transform (k = 1, 0, 0) gives the same out / bsum / ep_s1 / ep_s2 bit for bit -- squares 0..79 and the corner launch;
    (3) against the dz form (dz rounded to bf16 first) the outputs differ by that rounding only."""
    assert _lib.query("ka_conv3x3_dgrad_gated_supported", B, C, C, 1, 1)
    g = torch.Generator(device=DEV).manual_seed(B + C)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    du = torch.relu(rnd(B, 81, C)).to(torch.bfloat16)         # a masked gradient: exact zeros
    y, yprev = rnd(B, 81, C).to(torch.bfloat16), rnd(B, 81, C).to(torch.bfloat16)
    gate_add = torch.stack([torch.sigmoid(rnd(B, C)), 0.05 * rnd(B, C)]).contiguous()
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 1, 1, _lib.stream_ptr())
    k = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    mu, istd = 0.1 * rnd(C), torch.rand(C, device=DEV, generator=g) + 0.5

    def outs():
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        return nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16), nan(B, C), nan(B, C), nan(B, C)
    st = _lib.stream_ptr()
    out, dy, bsum, e1, e2 = outs()
    _lib.call("ka_conv3x3_dgrad_fused_gated", du, gate_add, y, k, dy, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
    torch.cuda.synchronize()
    for t in (out, dy, bsum, e1, e2):
        assert not bool(t.float().isnan().any())
    # (1) dy against the formula in float64, rounded once
    dz64 = du.double() * gate_add[0].double()[:, None, :] + gate_add[1].double()[:, None, :]
    dy64 = dz64 * k[:C].double() + k[C:2 * C].double() + y.double() * k[2 * C:].double()
    err = (dy.double() - dy64).abs()
    assert bool((err <= 2.0 ** -8 * dy64.abs() + 1e-5).all())          # half a bf16 ulp + the fp32 arithmetic of three fused steps
    assert float((dy.double() != dy64.float().to(torch.bfloat16).double()).double().mean()) < 2e-3   # (fp32 vs float64 before the one rounding)
    # (2) the plain kernel on that dy, identity transform
    ident = torch.cat([torch.ones(C, device=DEV), torch.zeros(2 * C, device=DEV)])
    out2, dy2, bsum2, e12, e22 = outs()
    _lib.call("ka_conv3x3_dgrad_fused", dy, y, ident, dy2, wp, out2, bsum2, yprev, sc, sh, mu, istd, e12, e22, B, C, C, 1, st)
    torch.cuda.synchronize()
    for name, a, b in (("dy", dy2, dy), ("out", out2, out), ("bsum", bsum2, bsum), ("ep_s1", e12, e1), ("ep_s2", e22, e2)):
        assert torch.equal(a, b), name
    # (3) the dz form: dz rounded to bf16 before the transform
    dz = dz64.float().to(torch.bfloat16)
    out3, dy3, bsum3, e13, e23 = outs()
    _lib.call("ka_conv3x3_dgrad_fused", dz, y, k, dy3, wp, out3, bsum3, yprev, sc, sh, mu, istd, e13, e23, B, C, C, 1, st)
    torch.cuda.synchronize()
    d = (dy3.float() - dy.float()).abs()
    assert bool((d <= 2.0 ** -6 * dy.float().abs() + 2.0 ** -7 * dz.float().abs() * k[:C].abs() + 1e-5).all())
    rel = float((out3.float() - out.float()).norm() / out.float().norm())
    assert rel < 4e-3, rel                                            # two roundings against one, through a 2304-term sum


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_pack_multi_equals_single_packs(dtn):
    """ka_pack_conv3x3_multi (all layers in one launch) writes byte-identical packs to per-layer ka_pack_conv3x3 calls."""
    dt = DT[dtn]
    code = _lib.dtype_code(dt)
    cpk = 32 if dt == torch.bfloat16 else 16
    g = torch.Generator().manual_seed(77)
    specs = [(64, 50, 64, 64, 0), (64, 64, 64, 64, 0), (64, 64, 64, 64, 1), (128, 64, 128, 64, 0), (128, 64, 64, 128, 1)]
    ws, singles, multis, jobs, mx = [], [], [], [], 0
    for co, ci, nout, kin, mode in specs:
        w = torch.randn(co, ci, 3, 3, generator=g).to(DEV)
        n = 9 * (kin // cpk) * (nout // 16) * 64
        a = torch.zeros(n * 16, dtype=torch.uint8, device=DEV); b = torch.zeros_like(a)
        _lib.call("ka_pack_conv3x3", w, a, co, ci, nout, kin, mode, code, st())
        jobs.append([w.data_ptr(), b.data_ptr(), co, ci, nout, kin, mode, 0]); mx = max(mx, n)
        ws.append(w); singles.append(a); multis.append(b)
    table = torch.tensor(jobs, dtype=torch.int64).to(DEV)
    _lib.call("ka_pack_conv3x3_multi", table, len(jobs), mx, code, st())
    torch.cuda.synchronize()
    for a, b in zip(singles, multis):
        assert torch.equal(a, b)


def test_bn_coefficients_from_partials_match_two_step_path():
    """ka_bn_reduce / ka_pair_reduce with sums == NULL + ka_bn_(bwd_)coeffs_parts == the reduced-sums path."""
    B, C = 37, 64
    g = torch.Generator().manual_seed(5)
    bsum = torch.randn(B, C, generator=g).to(DEV); sq = (torch.rand(B, C, generator=g) + 1.0).to(DEV) * 81
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    count = float(B * 81)
    ws = torch.empty(_lib.query("ka_reduce_workspace_doubles", C), dtype=torch.float64, device=DEV)
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        sc, sh, mu, istd = (torch.empty(C, device=DEV) for _ in range(4))
        if fused:
            _lib.call("ka_bn_reduce", bsum, B, sq, B, C, None, ws, st())
            _lib.call("ka_bn_coeffs_parts", ws, count, gamma, beta, rm, rv, nbt, 0.1, 1e-5, sc, sh, mu, istd, C, st())
        else:
            _lib.call("ka_bn_reduce", bsum, B, sq, B, C, sums, ws, st())
            _lib.call("ka_bn_coeffs", sums, count, None, gamma, beta, rm, rv, nbt, 0.1, 1e-5, sc, sh, mu, istd, C, st())
        outs.append((sc, sh, mu, istd, rm, rv, nbt.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    s1, s2 = torch.randn(B, C, generator=g).to(DEV), torch.randn(B, C, generator=g).to(DEV)
    mu, istd = outs[0][2], outs[0][3]
    outs = []
    for fused in (False, True):
        dg, db, k = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(3 * C, device=DEV)
        if fused:
            _lib.call("ka_pair_reduce", s1, s2, B, C, None, ws, st())
            _lib.call("ka_bn_bwd_coeffs_parts", ws, count, gamma, mu, istd, dg, db, k, C, 1, st())
        else:
            _lib.call("ka_pair_reduce", s1, s2, B, C, sums, ws, st())
            _lib.call("ka_bn_bwd_coeffs", sums, sums, count, None, gamma, mu, istd, dg, db, k, C, 1, st())
        outs.append((dg, db, k))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,R,C", [(37, 5, 64), (515, 129, 256), (4096, 256, 256), (300, 300, 96), (64, 3, 32)])
def test_bn_statistics_in_one_launch_equal_the_two_launches(B, R, C):
    """ka_bn_reduce_coeffs / ka_pair_reduce_bwd_coeffs (stage-1 reduce + coefficients by the last-arriving workgroup of a column
    group) == ka_bn_reduce / ka_pair_reduce (sums = NULL) + ka_*_coeffs_parts, every output bit for bit -- repeated back to back
    on changing inputs, so that a workgroup that read a partial row before it was published would show."""
    g = torch.Generator().manual_seed(B + C)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    count = float(B * 81)
    ws_a = torch.empty(_lib.query("ka_reduce_workspace_doubles", C), dtype=torch.float64, device=DEV)
    ws_b = torch.full_like(ws_a, float("nan"))
    cnt = torch.zeros(64, dtype=torch.int32, device=DEV)
    rm_a, rv_a, rm_b, rv_b = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt_a, nbt_b = torch.zeros((), dtype=torch.int64, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    for it in range(40):
        bsum = torch.randn(B, C, generator=g).to(DEV); sq = ((torch.rand(R, C, generator=g) + 1.0) * 81 * B / R).to(DEV)
        a = [torch.empty(C, device=DEV) for _ in range(4)]; b = [torch.full((C,), float("nan"), device=DEV) for _ in range(4)]
        _lib.call("ka_bn_reduce", bsum, B, sq, R, C, None, ws_a, st())
        _lib.call("ka_bn_coeffs_parts", ws_a, count, gamma, beta, rm_a, rv_a, nbt_a, 0.1, 1e-5, *a, C, st())
        _lib.call("ka_bn_reduce_coeffs", bsum, B, sq, R, C, ws_b, cnt, count, gamma, beta, rm_b, rv_b, nbt_b, 0.1, 1e-5, *b, st())
        s1, s2 = torch.randn(B, C, generator=g).to(DEV), torch.randn(B, C, generator=g).to(DEV)
        ka = [torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(3 * C, device=DEV)]
        kb = [torch.full_like(t, float("nan")) for t in ka]
        _lib.call("ka_pair_reduce", s1, s2, B, C, None, ws_a, st())
        _lib.call("ka_bn_bwd_coeffs_parts", ws_a, count, gamma, a[2], a[3], *ka, C, 1, st())
        _lib.call("ka_pair_reduce_bwd_coeffs", s1, s2, B, C, ws_b, cnt, count, gamma, b[2], b[3], *kb, 1, st())
        torch.cuda.synchronize()
        for name, x, y in zip(("scale", "shift", "mean", "invstd", "dgamma", "dbeta", "k"), a + ka, b + kb):
            assert torch.equal(x, y), (it, name)
        assert int(cnt.abs().sum()) == 0, "arrival counters must be back at zero after every launch"
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b) and int(nbt_a) == int(nbt_b) == 40


@pytest.mark.parametrize("dtn,B,cin,cout", [("f32", 1, 48, 48), ("f32", 2, 96, 80), ("bf16", 7, 96, 32), ("bf16", 1, 64, 96)])
def test_conv3x3_odd_shapes(dtn, B, cin, cout):
    """single boards, channel counts that are not powers of two (chunking by a divisor of Cin, a lone last output tile)"""
    dt = DT[dtn]
    g = torch.Generator().manual_seed(B * 31 + cin + cout)
    x = torch.randn(B, cin, 9, 9, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    ref = F.conv2d(rnd(x, dt), rnd(w, dt), padding=1)
    out, bsum, sq = run_conv(to_nhwc(x, dt), pack(w, dt, 0, cout, cin), B, cin, cout, dt)
    close(from_nhwc(out), ref, dt)
    close(bsum.cpu(), ref.sum(dim=(2, 3)), torch.float32, k=20 if dt == torch.float32 else 200)
    close(sq.sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3)), torch.float32, k=20 if dt == torch.float32 else 200)
    refd = torch.nn.grad.conv2d_input((B, cin, 9, 9), rnd(w, dt), rnd(ref, dt), padding=1)
    outd, _, _ = run_conv(to_nhwc(ref, dt), pack(w, dt, 1, cin, cout), B, cout, cin, dt)
    close(from_nhwc(outd), refd, dt)


@pytest.mark.parametrize("M,K1,ldx,H,N2,affine", [(37, 768, 1024, 128, 256, False), (130, 256, 256, 16, 512, True),
                                                   (5, 768, 1024, 256, 3, False), (16, 128, 128, 32, 1, True),
                                                   (4097, 64, 64, 64, 139, False)])
def test_fc_chain_matches_two_linears(M, K1, ldx, H, N2, affine):
    """y = W2 relu(W1 x' + b1) + b2 in one launch (global-pool bias, squeeze-excite and head FC chains) vs fp32 torch."""
    g = torch.Generator().manual_seed(M + K1 + H + N2)
    x = torch.randn(M, ldx, generator=g)
    W1, b1 = torch.randn(H, K1, generator=g) / K1 ** 0.5, torch.randn(H, generator=g)
    W2, b2 = torch.randn(N2, H, generator=g) / H ** 0.5, torch.randn(N2, generator=g)
    sc, sh, alpha = torch.randn(K1, generator=g), torch.randn(K1, generator=g), 1.0 / 81.0
    assert _lib.query("ka_fc_chain_supported", K1, ldx, H, N2) == 1
    xp_ref = sc * (x[:, :K1] * alpha) + sh if affine else x[:, :K1]
    hid_ref = torch.relu(xp_ref @ W1.t() + b1)
    y_ref = hid_ref @ W2.t() + b2
    y = torch.empty(M, N2, device=DEV)
    hid = torch.empty(M, H, device=DEV)
    xp = torch.empty(M, K1, device=DEV) if affine else None
    _lib.call("ka_fc_chain", x.to(DEV), sc.to(DEV) if affine else None, sh.to(DEV) if affine else None, alpha, W1.to(DEV),
              b1.to(DEV), W2.to(DEV), b2.to(DEV), xp, hid, y, M, K1, ldx, H, N2, st())
    assert torch.allclose(hid.cpu(), hid_ref, rtol=1e-5, atol=2e-5)
    assert torch.allclose(y.cpu(), y_ref, rtol=1e-5, atol=5e-5)
    if affine:
        assert torch.allclose(xp.cpu(), xp_ref, rtol=1e-6, atol=1e-6)
    # without the optional outputs and biases
    y2 = torch.empty(M, N2, device=DEV)
    _lib.call("ka_fc_chain", x.to(DEV), None, None, 1.0, W1.to(DEV), None, W2.to(DEV), None, None, None, y2, M, K1, ldx, H, N2, st())
    assert torch.allclose(y2.cpu(), torch.relu(x[:, :K1] @ W1.t()) @ W2.t(), rtol=1e-5, atol=5e-5)
    assert _lib.query("ka_fc_chain_supported", K1, ldx, 48, N2) == 0 and _lib.query("ka_fc_chain_supported", 100, 100, H, N2) == 0


def test_masked_softmax_for_action_selection():
    """probs over the legal actions + legal counts in one launch == the reference's masked_fill/softmax/Categorical chain
    (katago_ppo.py:566-584); torch.multinomial on them draws what Categorical(probs).sample() draws."""
    B, A = 9, 11259
    g = torch.Generator().manual_seed(2)
    logits = 4 * torch.randn(B, A, generator=g)
    legal = torch.rand(B, A, generator=g) < 0.02
    legal[:, 100] = True
    legal[3] = False                                   # a terminal-state row: count 0, row ignored by the caller
    ref = torch.softmax(logits.masked_fill(~legal, float("-inf")), dim=-1)
    probs = torch.empty(B, A, device=DEV)
    cnt = torch.empty(B, dtype=torch.int32, device=DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.call("ka_masked_softmax", logits.to(DEV), legal.to(DEV), probs, cnt, flags, B, A, 0, st())
    ok = [i for i in range(B) if i != 3]
    assert cnt.cpu().tolist() == legal.sum(-1).tolist() and flags.item() == 0
    assert torch.allclose(probs.cpu()[ok], ref[ok], rtol=2e-5, atol=1e-9)
    assert float(probs[ok].sum(-1).sub(1).abs().max()) < 1e-5 and bool((probs.cpu()[ok][~legal[ok]] == 0).all())
    bits = torch.empty(B, (A + 31) // 32, dtype=torch.int32, device=DEV)
    _lib.call("ka_pack_mask_bits", legal.to(DEV), bits, B, A, st())
    probs2 = torch.empty_like(probs)
    _lib.call("ka_masked_softmax", logits.to(DEV), bits, probs2, cnt, flags, B, A, (A + 31) // 32, st())
    assert torch.equal(probs2[ok], probs[ok])
    torch.manual_seed(7)
    a_ref = torch.distributions.Categorical(ref[ok].to(DEV), validate_args=False).sample()
    torch.manual_seed(7)
    a_got = torch.multinomial(probs[ok], 1, True).squeeze(1)
    assert torch.equal(a_ref, a_got)
    bad = logits.clone(); bad[5, 7] = float("nan")
    _lib.call("ka_masked_softmax", bad.to(DEV), legal.to(DEV), probs, cnt, flags, B, A, 0, st())
    assert flags.item() == 1


def test_grouped_fc_weight_gradients():
    """every FC weight / bias gradient of a backward pass in one launch == per-job dY^T X and column sums of dY."""
    g = torch.Generator().manual_seed(31)
    shapes = [(97, 128, 768, 1024, 0, True), (97, 256, 128, 128, 0, True), (97, 512, 16, 16, 0, True), (97, 16, 256, 256, 0, True),
              (97, 3, 40, 40, 0, False), (300, 70, 33, 36, 1, True)]            # (M, N, K, ldx, x_bf16, has_bias)
    rows, keep, wg = [], [], 0
    for M, N, K, ldx, xbf, has_b in shapes:
        dy = torch.randn(M, N, generator=g)
        x = torch.randn(M, ldx, generator=g)
        xd = x.bfloat16().to(DEV) if xbf else x.to(DEV)
        dW = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV) if has_b else None
        dyd = dy.to(DEV)
        keep.append((dy, x.bfloat16().float() if xbf else x, dW, db, dyd, xd, K))
        rows.append([dyd.data_ptr(), xd.data_ptr(), dW.data_ptr(), db.data_ptr() if has_b else 0, M, N, K, ldx, xbf, wg])
        wg += ((N + 63) // 64) * ((K + (1 if has_b else 0) + 63) // 64)
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    _lib.call("ka_gemm_grouped_wgrad", table, len(rows), wg, st())
    for dy, x, dW, db, _, _, K in keep:
        assert torch.allclose(dW.cpu(), dy.t() @ x[:, :K], rtol=1e-4, atol=1e-3)
        if db is not None:
            assert torch.allclose(db.cpu(), dy.sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B", [512, 515, 4096])
def test_five_row_tiles_plus_corner_equal_six_row_tiles(ka_env, B):
    """Training batches of the 256-channel tower compute squares 0..79 as five row tiles and square 80 of sixteen boards at a
    time in conv3x3_corner_kernel (default); KA_CONV_MT=6 is the six-row-tile form.  Same products in the same order: every
    output element bit for bit (all four launch kinds, both main kernels); the per-board sums add the corner's term last
    instead of inside a lane's partial -- equal up to fp32 re-association."""
    C = 256
    g = torch.Generator(device=DEV).manual_seed(B + 1)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, x2, yprev = (rnd(B, 81, C).to(torch.bfloat16) for _ in range(3))
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    gb = rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    mu, istd = 0.1 * rnd(C), torch.rand(C, device=DEV, generator=g) + 0.5

    def run(kind):
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        out, dyo = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16)
        bsum, sq, e1, e2 = nan(B, C), nan(B, C), nan(B, C), nan(B, C)
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
        if kind == 2: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
        torch.cuda.synchronize()
        return (out, dyo), (bsum, sq, e1, e2)

    ka_env.set("KA_CONV_PC2", "0")                           # (the two-board form sums in another order: its own test below)
    for pc in ("0", "3"):                                    # conv3x3_kernel / the producer-consumer kernel for every form
        ka_env.set("KA_CONV_P", pc)
        for kind in range(4):
            ka_env.set("KA_CONV_MT", "6")
            ref_o, ref_s = run(kind)
            ka_env.unset("KA_CONV_MT")
            got_o, got_s = run(kind)
            for a, b in zip(ref_o, got_o):
                assert bool(((a == b) | (a.isnan() & b.isnan())).all()), (pc, kind)
            assert not got_o[0].float().isnan().any()
            for a, b in zip(ref_s, got_s):
                if bool(a.isnan().all()):
                    assert bool(b.isnan().all())
                    continue
                assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-6, (pc, kind)


@pytest.mark.parametrize("B", [515, 1024, 4096])
def test_producer_consumer_conv_is_bit_identical_to_conv3x3_kernel(ka_env, B):
    """conv3x3_pc_kernel (staging waves + MFMA waves, the default for the forward forms at training batch sizes): the same
    summation order, weight packs and epilogues as conv3x3_kernel -- all four launch kinds agree bit for bit, including a
    board count that does not divide the 256 persistent workgroups and the headline minibatch (4096: 16 boards per
    workgroup, the transform-input form conv2 runs in the step)."""
    C = 256
    g = torch.Generator(device=DEV).manual_seed(B)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, x2, yprev = (rnd(B, 81, C).to(torch.bfloat16) for _ in range(3))
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    gb = rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    mu, istd = 0.1 * rnd(C), torch.rand(C, device=DEV, generator=g) + 0.5

    def run(kind):
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        out, dyo = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16)
        bsum, sq, e1, e2 = nan(B, C), nan(B, C), nan(B, C), nan(B, C)
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
        if kind == 2: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
        torch.cuda.synchronize()
        return out, dyo, bsum, sq, e1, e2

    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    ka_env.set("KA_CONV_PC2", "0")                           # (the two-board form sums in another order: its own test below)
    for kind in range(4):
        ka_env.set("KA_CONV_P", "0")
        ref = run(kind)
        ka_env.set("KA_CONV_P", "3")                  # every form through the producer / consumer kernel
        got = run(kind)
        assert all(same(a.float(), b.float()) for a, b in zip(ref, got)), kind
        assert not ref[0].float().isnan().any()


@pytest.mark.parametrize("B", [515, 1024])
def test_forward_conv_keeps_its_transformed_input_for_the_weight_gradient(B):
    """ka_conv3x3_fwd_keep: same output and sums as ka_conv3x3_fwd, and x_out = bf16(relu(x * scale + shift) + bias) -- the operand the
    convolution multiplied -- such that ka_conv3x3_wgrad on it as a plain input gives the weight gradient of the fused-input form
    (which repeats that transform per tile) bit for bit."""
    C = 256
    if not _lib.query("ka_conv3x3_fwd_keep_supported", B, C, C, 1):
        pytest.skip("shape not taken by the two-board kernel")
    g = torch.Generator(device=DEV).manual_seed(B + 3)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, dy = rnd(B, 81, C).to(torch.bfloat16), rnd(B, 81, C).to(torch.bfloat16)
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, st())
    sc, sh, gb = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1, rnd(B, C) * 0.1
    nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
    o0, b0, q0 = nan(B, 81, C, dt=torch.bfloat16), nan(B, C), nan(B, C)
    o1, b1, q1, xk = nan(B, 81, C, dt=torch.bfloat16), nan(B, C), nan(B, C), nan(B, 81, C, dt=torch.bfloat16)
    _lib.call("ka_conv3x3_fwd", x, wp, o0, sc, sh, gb, 1, b0, q0, B, C, C, 1, st())
    _lib.call("ka_conv3x3_fwd_keep", x, wp, o1, sc, sh, gb, 1, b1, q1, xk, B, C, C, 1, st())
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(b0, b1) and torch.equal(q0, q1)
    ref = (torch.relu(torch.addcmul(sh, x.float(), sc)) + gb[:, None, :]).to(torch.bfloat16)     # (fma, max, add: one rounding to bf16)
    assert float((xk.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())
    ns = _lib.query("ka_wgrad_splits", B, C, C, 0)
    slab = torch.empty(ns * 9 * C * C, device=DEV)
    dw_f, dw_p = nan(C, C, 3, 3), nan(C, C, 3, 3)
    _lib.call("ka_conv3x3_wgrad", dy, x, sc, sh, gb, 1, slab, dw_f, B, C, C, C, 0, 0, 1, st())
    _lib.call("ka_conv3x3_wgrad", dy, xk, None, None, None, 0, slab, dw_p, B, C, C, C, 0, 0, 1, st())
    torch.cuda.synchronize()
    assert not bool(dw_f.isnan().any()) and torch.equal(dw_f, dw_p)


@pytest.mark.parametrize("B", [512, 515, 1024, 4096, 4302, 9000])
def test_in_kernel_corner_equals_the_corner_launch(ka_env, B):
    """KA_CONV_CORNER_IN: square 80 of up to eight board pairs as one more row tile inside conv3x3_pc2_kernel (rows left in the side
    buffer by the staging waves) == conv3x3_corner_kernel launched behind it: outputs, the written-back dy and every per-board sum bit
    for bit, all four launch kinds.  Board counts: a half-empty last pair (515), exactly eight pairs per workgroup (4096), nine in
    some (4302: a second, one-pair group and the ninth side slot), more than two groups (9000)."""
    C = 256
    g = torch.Generator(device=DEV).manual_seed(B + 11)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, x2, yprev = (rnd(B, 81, C).to(torch.bfloat16) for _ in range(3))
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    gb = rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    mu, istd = 0.1 * rnd(C), torch.rand(C, device=DEV, generator=g) + 0.5

    def run(kind):
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        out, dyo = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16)
        bsum, sq, e1, e2 = nan(B, C), nan(B, C), nan(B, C), nan(B, C)
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
        if kind == 2: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
        torch.cuda.synchronize()
        return out, dyo, bsum, sq, e1, e2

    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    for kind in range(4):
        ka_env.set("KA_CONV_CORNER_IN", "0")
        ref = run(kind)
        ka_env.set("KA_CONV_CORNER_IN", "1")                  # (the default; the masked form, kind 2, keeps the launch either way)
        got = run(kind)
        assert not bool(ref[0].float().isnan().any()), kind
        for name, a, b in zip(("out", "dy", "bsum", "sqpart", "ep_s1", "ep_s2"), ref, got):
            assert same(a.float(), b.float()), (kind, name)


@pytest.mark.parametrize("B,C", [(515, 256), (1024, 256), (4096, 256), (515, 128), (2048, 128)])
def test_two_board_conv_equals_the_one_board_forms_up_to_reassociation(ka_env, B, C):
    """conv3x3_pc2_kernel (two boards per weight fragment: the default for the forward forms and the plain-epilogue data gradient at
    training batch sizes) sums the k-steps of an output element in the order (64-channel chunk, tap, k-step) where the other
    kernels use (128-channel chunk, tap, k-step): the fp32 accumulators agree up to re-association, so the bf16 outputs are
    equal or -- rarely -- one ulp apart; the written-back dy (the staging transform) is bit-identical; per-board sums agree to
    1e-5.  All four launch kinds, a board count that leaves the last pair half empty (515), and both forms against an fp32
    convolution of the same bf16 operands (the yardstick conv3x3_kernel itself is held to).  C = 128 (BASELINE configs[1] at its
    minibatch 2048; keisei-ddp.toml's tower): four MFMA waves, two chunks, all 81 squares as six row tiles per board, no corner
    launch -- against conv3x3_kernel, which those shapes ran on before."""
    g = torch.Generator(device=DEV).manual_seed(B + 7)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, x2, yprev = (rnd(B, 81, C).to(torch.bfloat16) for _ in range(3))
    w = rnd(C, C, 3, 3) / 48
    wp = torch.empty(9 * (C // 32) * (C // 16) * 1024, dtype=torch.uint8, device=DEV)
    _lib.call("ka_pack_conv3x3", w, wp, C, C, C, C, 0, 1, _lib.stream_ptr())
    sc, sh = torch.rand(C, device=DEV, generator=g) + 0.5, rnd(C) * 0.1
    gb = rnd(B, C) * 0.1
    k3 = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, 0.1 * rnd(C), 0.2 * rnd(C)])
    mu, istd = 0.1 * rnd(C), torch.rand(C, device=DEV, generator=g) + 0.5

    def run(kind):
        nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), device=DEV).to(dt)
        out, dyo = nan(B, 81, C, dt=torch.bfloat16), nan(B, 81, C, dt=torch.bfloat16)
        bsum, sq, e1, e2 = nan(B, C), nan(B, C), nan(B, C), nan(B, C)
        st = _lib.stream_ptr()
        if kind == 0: _lib.call("ka_conv3x3_fwd", x, wp, out, None, None, None, 0, bsum, sq, B, C, C, 1, st)
        if kind == 1: _lib.call("ka_conv3x3_fwd", x, wp, out, sc, sh, gb, 1, bsum, sq, B, C, C, 1, st)
        if kind == 2: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, bsum, yprev, sc, sh, mu, istd, e1, e2, B, C, C, 1, st)
        if kind == 3: _lib.call("ka_conv3x3_dgrad_fused", x, x2, k3, dyo, wp, out, None, None, None, None, None, None, None, None, B, C, C, 1, st)
        torch.cuda.synchronize()
        return (out, dyo), (bsum, sq, e1, e2)

    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    for kind in range(4):
        ka_env.set("KA_CONV_PC2", "0")
        (o0, d0), s0 = run(kind)
        ka_env.set("KA_CONV_PC2", "3")                       # every form through the two-board kernel
        (o1, d1), s1 = run(kind)
        a, b = o0.float(), o1.float()
        assert not bool(a.isnan().any()) and not bool(b.isnan().any()), kind
        diff = (a - b).abs()
        assert bool((diff <= 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 2e-5).all()), kind      # one bf16 ulp (absolute floor near zero)
        assert float((diff != 0).float().mean()) < 0.01, kind                                       # measured: 1e-4 .. 3e-4 of the elements
        assert same(d0.float(), d1.float()), kind
        for u, v in zip(s0, s1):
            if bool(u.isnan().all()):
                assert bool(v.isnan().all()), kind
                continue
            tol = 2e-3 if kind == 2 else 1e-5                # (masked sums add the one-ulp output differences of up to 81 squares)
            assert float((u - v).abs().max()) <= tol * float(u.abs().max()) + 1e-6, kind
    n = 64
    xf = x[:n].float().reshape(n, 9, 9, C).permute(0, 3, 1, 2)
    ref32 = torch.nn.functional.conv2d(xf, w.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1).reshape(n, 81, C)
    errs = []
    for form in ("0", "3"):
        ka_env.set("KA_CONV_PC2", form)
        (out, _), _ = run(0)
        errs.append(float((out[:n].float() - ref32).abs().max()) / float(ref32.abs().max()))
    assert max(errs) < 6e-3 and abs(errs[0] - errs[1]) < 1e-3, errs       # both at the bf16 output rounding (3.9e-3)
