"""GPU: the TransformerModel HIP path (BASELINE config 5; reference transformer.py:37-95) -- kernels against plain fp32
CPU math, the model against reference-generated fixtures (g9: fp32 outputs, fp64 gradients), the bf16 mode against a
measured bound, and the statistics / reproducibility of the counter-based dropout."""
import math

import pytest
import torch

from keisei_amd import _lib
from keisei_amd.training.model_registry import build_model
from oracle import keisei_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def st():
    return _lib.stream_ptr(torch.device(DEV))


@pytest.mark.parametrize("M,N,K,split", [(300, 200, 64, 1), (130, 139, 96, 1), (257, 384, 2592, 1), (64, 96, 4096, 8)])
def test_gemm_nt_bf16(M, N, K, split):
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = torch.randn(N, K, generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).bfloat16()
    ref = a.float() @ b.float().T
    ad, bd = a.to(DEV), b.to(DEV)
    if split == 1:
        out = torch.empty(M, N, device=DEV)
        _lib.call("ka_tf_gemm_nt", ad, bd, out, bias.to(DEV), None, M, N, K, K, K, N, 0, 1, 1, 0.0, 0, st())
        want = torch.relu(ref + bias)
        assert torch.allclose(out.cpu(), want, rtol=1e-4, atol=1e-3 * math.sqrt(K) / 8)
        out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        _lib.call("ka_tf_gemm_nt", ad, bd, out16, bias.to(DEV), res.to(DEV), M, N, K, K, K, N, 1, 0, 1, 0.0, 0, st())
        want = (ref + bias + res.float()).bfloat16().float()
        assert float((out16.float().cpu() - want).abs().max()) <= 0.02 * float(want.abs().max())
    else:
        ns = _lib.query("ka_tf_gemm_nt_slabs", K, split)
        slab = torch.empty(ns, M, N, device=DEV)
        _lib.call("ka_tf_gemm_nt", ad, bd, slab, None, None, M, N, K, K, K, N, 0, 0, split, 0.0, 0, st())
        assert torch.allclose(slab.sum(0).cpu(), ref, rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("M,N,K,ldb,split", [(300, 200, 64, 64, 1), (130, 136, 96, 104, 1), (4097, 256, 1024, 1024, 7),
                                             (64, 8, 8, 8, 1), (1000, 768, 256, 256, 16)])
def test_gemm_tn_bf16(M, N, K, ldb, split):
    """dW = dY^T X from the row-major operands (LDS transpose reads): ragged token counts, column counts that do not
    fill a tile, padded operand rows, one slab and several."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, N, generator=g).bfloat16()
    b = torch.randn(M, ldb, generator=g).bfloat16()
    ref = a.float().T @ b.float()[:, :K]
    ns = _lib.query("ka_tf_gemm_tn_slabs", M, split)
    slab = torch.full((ns, N, K), float("nan"), device=DEV)
    _lib.call("ka_tf_gemm_tn", a.to(DEV), b.to(DEV), slab, M, N, K, N, ldb, K, split, st())
    torch.cuda.synchronize()
    assert ns >= 1 and (split == 1) == (ns == 1)
    assert torch.allclose(slab.sum(0).cpu(), ref, rtol=1e-4, atol=1e-3 * math.sqrt(M) / 4)
    # the same launch with the bias gradient riding along: identical slabs, column sums of A per token range
    slab2 = torch.full((ns, N, K), float("nan"), device=DEV); cs = torch.full((ns, N), float("nan"), device=DEV)
    _lib.call("ka_tf_gemm_tn_bias", a.to(DEV), b.to(DEV), slab2, cs, M, N, K, N, ldb, K, split, st())
    torch.cuda.synchronize()
    assert torch.equal(slab2, slab)
    assert torch.allclose(cs.sum(0).cpu(), a.float().sum(0), rtol=1e-5, atol=1e-3 * math.sqrt(M) / 4)


@pytest.mark.parametrize("M,N", [(81, 256), (300, 1024), (4096, 768), (1000, 272), (129, 64)])
def test_gemm_nt_k256_form_equals_the_tiled_kernel(ka_env, M, N):
    """K = 256 with a bf16 output takes the activation-stationary kernel (gemm_nt_k256_kernel); KA_TF_K256=0 sends the same call to
    gemm_nt_bf16_kernel.  Same products in the same k order, same epilogue statements: every epilogue form bit for bit
    (bias + ReLU + dropout, bias + dropout + residual, the masked input-gradient form), ragged row counts, partial last chunk."""
    K = 256
    g = torch.Generator().manual_seed(M * 3 + N)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    b = (torch.randn(N, K, generator=g) / 16).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    act = torch.randn(M, N, generator=g).bfloat16().to(DEV)

    def run():
        outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(4)]
        _lib.call("ka_tf_gemm_nt", a, b, outs[0], bias, None, M, N, K, K, K, N, 1, 1, 1, 0.1, 777, st())
        _lib.call("ka_tf_gemm_nt", a, b, outs[1], bias, res, M, N, K, K, K, N, 1, 0, 1, 0.2, 4242, st())
        _lib.call("ka_tf_gemm_nt", a, b, outs[2], None, None, M, N, K, K, K, N, 1, 0, 1, 0.0, 0, st())
        _lib.call("ka_tf_gemm_nt_masked", a, b, outs[3], act, M, N, K, K, K, N, 0.1, 99, st())
        torch.cuda.synchronize()
        return outs
    new = run()
    ka_env.set("KA_TF_K256", "0")
    old = run()
    for k, (x, y) in enumerate(zip(new, old)):
        assert bool(torch.isfinite(x.float()).all()) and torch.equal(x, y), k
    want = (a.float() @ b.float().T).cpu()
    assert float((new[2].float().cpu() - want).abs().max()) <= 0.02 * float(want.abs().max())


@pytest.mark.parametrize("M,N,K", [(1100, 1300, 1024), (1024, 1027, 1088), (1281, 2048, 1024), (2048, 1536, 4096)])
def test_gemm_nt_big_tile_form_equals_the_tiled_kernel(ka_env, M, N, K):
    """M, N, K >= 1024 with a bias-only epilogue (the policy layer's three products) take gemm_nt_big_kernel (256 x 256 tiles, an XCD
    walking 4 x 8 super-tiles, operands by LDS-DMA three k-tiles ahead under counted waits); KA_TF_BIG=0 sends the call to gemm_nt_bf16_kernel, whose grid for these shapes is the 8 x 8
    super-tile map, and KA_TF_MAP2D=0 to its one-m-tile-per-XCD map.  Same products in the same k order: bit for bit, fp32
    and bf16 outputs, ragged M and N (N not a multiple of 4: the scalar store path), and against fp32 math."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    b = (torch.randn(N, K, generator=g) / 32).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)

    def run():
        o32 = torch.full((M, N), float("nan"), device=DEV)
        _lib.call("ka_tf_gemm_nt", a, b, o32, bias, None, M, N, K, K, K, N, 0, 0, 1, 0.0, 0, st())
        ldc = (N + 7) // 8 * 8
        o16 = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.call("ka_tf_gemm_nt", a, b, o16, None, None, M, N, K, K, K, ldc, 1, 0, 1, 0.0, 0, st())
        torch.cuda.synchronize()
        return o32, o16[:, :N].clone()
    big = run()
    ka_env.set("KA_TF_BIG", "0")
    tiled = run()
    ka_env.set("KA_TF_MAP2D", "0")
    plain = run()
    for x, y, z in zip(big, tiled, plain):
        assert bool(torch.isfinite(x.float()).all()) and torch.equal(x, y) and torch.equal(y, z)
    want = (a.float() @ b.float().T + bias).cpu()
    assert torch.allclose(big[0].cpu(), want, rtol=1e-4, atol=1e-3 * math.sqrt(K) / 8)


@pytest.mark.parametrize("M,N,K,p", [(300, 256, 96, 0.1), (130, 200, 64, 0.0), (257, 1024, 256, 0.3)])
def test_masked_gemm_equals_gemm_then_dropout_relu_backward(M, N, K, p):
    """ka_tf_gemm_nt_masked (dropout + ReLU backward in the epilogue of the input-gradient GEMM) against the two-launch
    form it replaces: same (seed, element) mask, same activation test; full tiles take the LDS epilogue, ragged ones not."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    b = torch.randn(N, K, generator=g).bfloat16().to(DEV)
    act = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    seed = 424242
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_tf_gemm_nt", a, b, ref, None, None, M, N, K, K, K, N, 1, 0, 1, 0.0, 0, st())
    _lib.call("ka_tf_drop_apply", ref, act, None, ref, ref.numel(), p, seed, _lib.DTYPE_BF16, st())
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_tf_gemm_nt_masked", a, b, out, act, M, N, K, K, K, N, p, seed, st())
    torch.cuda.synchronize()
    r, o = ref.float().cpu(), out.float().cpu()
    assert torch.equal(r == 0, o == 0) or float(((r == 0) != (o == 0)).float().mean()) < 1e-4      # same mask
    assert float((r - o).abs().max()) <= 2e-2 * float(r.abs().max())


@pytest.mark.parametrize("dtn,M,d", [("bf16", 500, 256), ("f32", 130, 64), ("bf16", 77, 24)])
def test_layernorm_backward_second_output_is_the_dropped_gradient(dtn, M, d):
    """ka_tf_layernorm_bwd_drop: dx as ka_tf_layernorm_bwd writes it, and dx_drop == ka_tf_drop_apply(dx) for the same
    (seed, element) mask -- the 16-byte kernel (d = 256 bf16, d = 64 fp32) and the one-wave-per-row form (d = 24)."""
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    code = _lib.dtype_code(dt)
    g = torch.Generator().manual_seed(M + d)
    dy = torch.randn(M, d, generator=g).to(dt).to(DEV); x = torch.randn(M, d, generator=g).to(dt).to(DEV)
    dres = torch.randn(M, d, generator=g).to(dt).to(DEV)
    gam = (torch.rand(d, generator=g) + 0.5).to(DEV)
    mu = x.float().mean(1).contiguous(); rs = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    nparts = _lib.query("ka_tf_layernorm_parts", M)
    part = torch.empty((nparts + 1) * 2 * d, device=DEV)
    p, seed = 0.2, 991
    outs = []
    for fused in (False, True):
        dx = torch.full((M, d), float("nan"), dtype=dt, device=DEV); dxd = torch.full((M, d), float("nan"), dtype=dt, device=DEV)
        dg = torch.empty(d, device=DEV); db = torch.empty(d, device=DEV)
        if fused:
            _lib.call("ka_tf_layernorm_bwd_drop", dy, x, gam, mu, rs, dres, dx, dxd, p, seed, part, dg, db, M, d, code, st())
        else:
            _lib.call("ka_tf_layernorm_bwd", dy, x, gam, mu, rs, dres, dx, part, dg, db, M, d, code, st())
            _lib.call("ka_tf_drop_apply", dx, None, None, dxd, dx.numel(), p, seed, code, st())
        torch.cuda.synchronize()
        outs.append((dx.float().cpu(), dxd.float().cpu(), dg.cpu(), db.cpu()))
    (dx0, dd0, dg0, db0), (dx1, dd1, dg1, db1) = outs
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dd0 == 0, dd1 == 0)
    assert float((dd0 - dd1).abs().max()) <= (1e-6 if dt == torch.float32 else 1e-2) * float(dd0.abs().max())
    kept = float((dd1 != 0).float().mean())
    assert abs(kept - (1 - p)) < 0.03


def test_transpose_and_cast_pad():
    x = torch.randn(70, 45)
    out = torch.full((45, 96), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_tf_transpose_pad", x.to(DEV), out, 70, 45, 45, 96, _lib.DTYPE_F32, st())
    assert torch.equal(out[:, :70].cpu(), x.T.bfloat16()) and float(out[:, 70:].abs().max()) == 0
    out = torch.full((70, 64), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_tf_cast_pad", x.bfloat16().to(DEV), out, 70, 45, 45, 64, _lib.DTYPE_BF16, st())
    assert torch.equal(out[:, :45].cpu(), x.bfloat16()) and float(out[:, 45:].abs().max()) == 0


def test_weight_cache_in_one_launch_equals_the_per_layer_copies():
    """ka_tf_weights16_multi (both bf16 copies of every layer from one read of the weight) == ka_tf_cast_pad + ka_tf_transpose_pad per
    layer, pad columns included; shapes with ragged 64-tiles and pad columns on both copies.  ka_reduce_slabs2 == two ka_reduce_slabs."""
    g = torch.Generator().manual_seed(3)
    r32 = lambda v: (v + 31) // 32 * 32
    shapes = [(768, 256), (256, 256), (1024, 256), (256, 1024), (70, 44), (139, 2592), (11, 4), (256, 50), (33, 7)]
    ws = [torch.randn(N, K, generator=g).to(DEV) for N, K in shapes]
    ref, got, rows, first = [], [], [], 0
    for w in ws:
        N, K = w.shape
        a = torch.full((N, r32(K)), 7.0, dtype=torch.bfloat16, device=DEV); b = torch.full((K, r32(N)), 7.0, dtype=torch.bfloat16, device=DEV)
        _lib.call("ka_tf_cast_pad", w, a, N, K, K, r32(K), _lib.DTYPE_F32, st())
        _lib.call("ka_tf_transpose_pad", w, b, N, K, K, r32(N), _lib.DTYPE_F32, st())
        ref.append((a, b))
        c = torch.full_like(a, 9.0); d = torch.full_like(b, 9.0)
        got.append((c, d))
        rows.append([w.data_ptr(), c.data_ptr(), d.data_ptr(), N, K, r32(K), r32(N), first])
        first += ((r32(N) + 63) // 64) * ((r32(K) + 63) // 64)
    _lib.call("ka_tf_weights16_multi", torch.tensor(rows, dtype=torch.int64).to(DEV), len(rows), first, st())
    torch.cuda.synchronize()
    for (a, b), (c, d), shp in zip(ref, got, shapes):
        assert torch.equal(a, c) and torch.equal(b, d), shp
    ns, na, nb = 7, 1000, 37
    sa, sb = torch.randn(ns, na, generator=g).to(DEV), torch.randn(ns, nb, generator=g).to(DEV)
    oa, ob, pa, pb = (torch.empty(n, device=DEV) for n in (na, nb, na, nb))
    _lib.call("ka_reduce_slabs", sa, oa, ns, na, 0, st()); _lib.call("ka_reduce_slabs", sb, ob, ns, nb, 0, st())
    _lib.call("ka_reduce_slabs2", sa, pa, na, sb, pb, nb, ns, st())
    torch.cuda.synchronize()
    assert torch.equal(oa, pa) and torch.equal(ob, pb)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm_forward_backward(dt):
    M, d = 243, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, d, generator=g).to(dt)
    gam, bet = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    dy = torch.randn(M, d, generator=g).to(dt)
    dres = torch.randn(M, d, generator=g).to(dt)
    xr = x.float().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    y.backward(dy.float())
    code = _lib.dtype_code(dt)
    xd = x.to(DEV)
    yd = torch.empty_like(xd); mu = torch.empty(M, device=DEV); rs = torch.empty(M, device=DEV)
    _lib.call("ka_tf_layernorm_fwd", xd, gam.to(DEV), bet.to(DEV), yd, mu, rs, M, d, 1e-5, code, st())
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert torch.allclose(yd.float().cpu(), y.detach(), rtol=tol, atol=tol)
    nparts = _lib.query("ka_tf_layernorm_parts", M)
    ws = torch.empty((nparts + 1) * 2 * d, device=DEV)
    dx = torch.empty_like(xd); dg = torch.empty(d, device=DEV); db = torch.empty(d, device=DEV)
    _lib.call("ka_tf_layernorm_bwd", dy.to(DEV), xd, gam.to(DEV), mu, rs, dres.to(DEV), dx, ws, dg, db, M, d, code, st())
    assert torch.allclose(dx.float().cpu(), xr.grad + dres.float(), rtol=tol, atol=2 * tol)
    assert torch.allclose(dg.cpu(), gr.grad, rtol=1e-3, atol=1e-3 if dt == torch.float32 else 0.3)
    assert torch.allclose(db.cpu(), br.grad, rtol=1e-3, atol=1e-3 if dt == torch.float32 else 0.3)


def _attn_ref(qkv, B, H, dh):
    d = H * dh
    q, k, v = (t.reshape(B, 81, H, dh).transpose(1, 2) for t in qkv.reshape(B, 81, 3, d).unbind(2))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B * 81, d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dt,H,dh", [(torch.float32, 4, 8), (torch.float32, 2, 32), (torch.bfloat16, 8, 32), (torch.bfloat16, 2, 64),
                                     (torch.bfloat16, 4, 8), (torch.float32, 3, 24)])
def test_attention_forward_backward(dt, H, dh):
    B, d = 3, H * dh
    g = torch.Generator().manual_seed(H * 100 + dh)
    qkv = torch.randn(B * 81, 3 * d, generator=g).to(dt)
    dout = torch.randn(B * 81, d, generator=g).to(dt)
    ref_in = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(ref_in, B, H, dh)
    o_ref.backward(dout.double())
    code = _lib.dtype_code(dt)
    qd = qkv.to(DEV)
    out = torch.empty(B * 81, d, dtype=dt, device=DEV); lse = torch.empty(B, H, 81, device=DEV)
    _lib.call("ka_tf_attention_fwd", qd, out, lse, B, H, dh, 0.0, 0, code, st())
    tol = 2e-5 if dt == torch.float32 else 3e-2
    assert torch.allclose(out.double().cpu(), o_ref.detach(), rtol=tol, atol=tol)
    assert torch.allclose(lse.double().cpu(), lse_ref.detach(), rtol=1e-3 if dt == torch.bfloat16 else 1e-5, atol=tol)
    dq = torch.full((B * 81, 3 * d), float("nan"), dtype=dt, device=DEV)
    _lib.call("ka_tf_attention_bwd", qd, dout.to(DEV), lse, dq, B, H, dh, 0.0, 0, code, st())
    ref = ref_in.grad
    err = float((dq.double().cpu() - ref).norm() / ref.norm())
    assert err < (2e-5 if dt == torch.float32 else 2e-2), err


@pytest.mark.parametrize("H,dh,p", [(8, 32, 0.1), (4, 8, 0.3), (3, 16, 0.0)])
def test_register_attention_equals_the_lds_form(ka_env, H, dh, p):
    """bf16, head dimension <= 32: the register-resident kernels (the default) against the LDS-tile kernels
    (KA_TF_ATTN_LDS=1) on the same seed -- same dropout mask in both score orientations of the backward, same softmax."""
    B, d, seed = 5, H * dh, 987654321
    g = torch.Generator().manual_seed(H + dh)
    qkv = torch.randn(B * 81, 3 * d, generator=g).bfloat16().to(DEV)
    dout = torch.randn(B * 81, d, generator=g).bfloat16().to(DEV)
    res = {}
    for form in ("1", ""):
        if form:
            ka_env.set("KA_TF_ATTN_LDS", form)
        else:
            ka_env.unset("KA_TF_ATTN_LDS")
        out = torch.full((B * 81, d), float("nan"), dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, H, 81, device=DEV)
        _lib.call("ka_tf_attention_fwd", qkv, out, lse, B, H, dh, p, seed, _lib.DTYPE_BF16, st())
        dq = torch.full((B * 81, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.call("ka_tf_attention_bwd", qkv, dout, lse, dq, B, H, dh, p, seed, _lib.DTYPE_BF16, st())
        torch.cuda.synchronize()
        res[form] = (out.float().cpu(), lse.cpu(), dq.float().cpu())
    (o0, l0, g0), (o1, l1, g1) = res["1"], res[""]
    assert torch.allclose(l1, l0, rtol=1e-5, atol=1e-5)
    assert float((o1 - o0).abs().max()) <= 2e-2 * float(o0.abs().max())
    assert float((g1 - g0).norm() / g0.norm()) < 1e-2
    # ... and the two-launch backward (dQ | dK, dV with D = rowsum(dout * out) from the forward's bf16 output) against both
    out = torch.empty(B * 81, d, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, H, 81, device=DEV)
    _lib.call("ka_tf_attention_fwd", qkv, out, lse, B, H, dh, p, seed, _lib.DTYPE_BF16, st())
    dq2 = torch.full((B * 81, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.call("ka_tf_attention_bwd_o", qkv, out, dout, lse, dq2, B, H, dh, p, seed, _lib.DTYPE_BF16, st())
    torch.cuda.synchronize()
    g2 = dq2.float().cpu()
    assert bool(torch.isfinite(g2).all())
    assert float((g2 - g1).norm() / g1.norm()) < 1e-2 and float((g2 - g0).norm() / g0.norm()) < 1.5e-2
    for blk in range(3):                                     # dQ, dK, dV separately
        a, b_ = g2[:, blk * d:(blk + 1) * d], g1[:, blk * d:(blk + 1) * d]
        assert float((a - b_).norm() / b_.norm()) < 1.5e-2, blk


def test_attention_dropout_is_consistent_between_forward_and_backward():
    """With dropout the kernel's own forward is the reference for its backward: d(sum(out * w)) / d(qkv) by central
    differences of the fp32 forward (same seed = same mask) against the backward kernel; and the mask keeps 1 - p."""
    B, H, dh, p, seed = 1, 2, 16, 0.3, 1234567
    d = H * dh
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(81, 3 * d, generator=g).to(DEV)
    w = torch.randn(81, d, generator=g).to(DEV)

    def fwd(x):
        out = torch.empty(81, d, device=DEV); lse = torch.empty(B, H, 81, device=DEV)
        _lib.call("ka_tf_attention_fwd", x.contiguous(), out, lse, B, H, dh, p, seed, _lib.DTYPE_F32, st())
        return out, lse

    out, lse = fwd(qkv)
    dq = torch.empty_like(qkv)
    _lib.call("ka_tf_attention_bwd", qkv, w, lse, dq, B, H, dh, p, seed, _lib.DTYPE_F32, st())
    idx = [(0, 0), (5, 3), (40, d + 7), (80, 2 * d + 1), (17, 2 * d + dh + 2)]
    for (r, c) in idx:
        e = torch.zeros_like(qkv); e[r, c] = 1e-2
        num = float((((fwd(qkv + e)[0] - fwd(qkv - e)[0]) * w).sum() / 2e-2))
        assert abs(num - float(dq[r, c])) <= 2e-2 * max(1.0, abs(num)), (r, c, num, float(dq[r, c]))
    out0, _ = fwd(qkv)
    assert torch.equal(out, out0)                       # same seed, same mask
    # v = 1: out = sum of kept probabilities / (1 - p) -> mean 1 over rows when the mask keeps 1 - p of the mass
    ones = qkv.clone(); ones[:, 2 * d:] = 1.0
    kept = fwd(ones)[0]
    assert abs(float(kept.mean()) - 1.0) < 0.05


def _fixture_model(tag, p):
    m = build_model("transformer", p)
    m.load_state_dict(orc.hash_fill(m.state_dict()), strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m.to(DEV)


# the third one is the head shape bench.py's transformer workload runs (VERDICT r2 "What's weak" 4): it exercises the d = 256
# kernels at model level -- the activation-stationary K = 256 GEMM, 8 heads of 32, the 16-byte LayerNorm -- on two boards
CONFIGS = [("d32h4L2.", {"d_model": 32, "nhead": 4, "num_layers": 2}), ("d64h2L1.", {"d_model": 64, "nhead": 2, "num_layers": 1}),
           ("d256h8L1.", {"d_model": 256, "nhead": 8, "num_layers": 1})]
FIXTURE = {"d32h4L2.": "g9_transformer", "d64h2L1.": "g9_transformer", "d256h8L1.": "g9_transformer_d256"}


@pytest.mark.parametrize("tag,p", CONFIGS)
def test_transformer_fp32_matches_reference(golden, tag, p):
    g = golden(FIXTURE[tag])
    m = _fixture_model(tag, p)
    obs = g[tag + "obs"].to(DEV)
    B = obs.shape[0]
    m.eval()
    with torch.no_grad():
        pol, val = m(obs)
    assert pol.shape == (B, 11259) and val.shape == (B, 1)
    assert torch.allclose(pol.cpu(), g[tag + "eval.policy"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(val.cpu(), g[tag + "eval.value"], rtol=1e-4, atol=2e-5)
    m.train()
    pol, val = m(obs)
    assert torch.allclose(pol.detach().cpu(), g[tag + "train.policy"], rtol=1e-4, atol=2e-5)
    cp = orc._hash_uniform(B * 11259, 7101).float().reshape(B, 11259).to(DEV)
    cv = orc._hash_uniform(B, 7102).float().reshape(B, 1).to(DEV)
    ((pol * cp).sum() / B + (val * cv).sum()).backward()
    names = list(g.np(tag + "grad_names"))
    norms = dict(zip(names, g.np(tag + "grad_norms64")))
    worst_n = worst_l2 = 0.0
    for n, prm in m.named_parameters():
        assert prm.grad is not None, n
        worst_n = max(worst_n, abs(float(prm.grad.double().norm()) - norms[n]) / (norms[n] + 1e-30))
        for key, got in ((f"{tag}grad64.{n}", prm.grad), (f"{tag}grad64.{n}[:8]", prm.grad[:8])):
            if key in g:
                ref = g[key].double()
                worst_l2 = max(worst_l2, float((got.double().cpu() - ref).norm() / (ref.norm() + 1e-30)))
    print(f"{tag} fp32 gradients vs reference fp64: norm {worst_n:.2e}, rel L2 {worst_l2:.2e}")
    assert worst_n < 1e-4 and worst_l2 < 1e-4


@pytest.mark.parametrize("tag,p", CONFIGS)
def test_transformer_bf16_bound(golden, tag, p):
    """bf16 autocast: policy logits within 3 % of |logit|max of the reference's fp32 output (measured ~1 %), gradient
    norms within 10 %, every stored gradient tensor within 15 % relative L2 of the reference's fp64 gradient."""
    g = golden(FIXTURE[tag])
    m = _fixture_model(tag, p)
    obs = g[tag + "obs"].to(DEV)
    B = obs.shape[0]
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pol, val = m(obs)
    ref = g[tag + "train.policy"]
    e = float((pol.detach().float().cpu() - ref).abs().max()) / float(ref.abs().max())
    cp = orc._hash_uniform(B * 11259, 7101).float().reshape(B, 11259).to(DEV)
    cv = orc._hash_uniform(B, 7102).float().reshape(B, 1).to(DEV)
    ((pol.float() * cp).sum() / B + (val.float() * cv).sum()).backward()
    names = list(g.np(tag + "grad_names"))
    norms = dict(zip(names, g.np(tag + "grad_norms64")))
    worst_n = worst_l2 = 0.0
    for n, prm in m.named_parameters():
        worst_n = max(worst_n, abs(float(prm.grad.double().norm()) - norms[n]) / (norms[n] + 1e-30))
        key = f"{tag}grad64.{n}"
        if key in g:
            refg = g[key].double()
            worst_l2 = max(worst_l2, float((prm.grad.double().cpu() - refg).norm() / (refg.norm() + 1e-30)))
    print(f"{tag} bf16: policy {e:.4f} of |logit|max, gradient norms off by {worst_n:.3f}, rel L2 {worst_l2:.3f}")
    assert e < 0.03 and worst_n < 0.10 and worst_l2 < 0.15


def test_transformer_training_mode_dropout():
    """Train mode with the reference's dropout 0.1: outputs differ from eval mode and from call to call (fresh seed per
    forward), stay finite, and every parameter receives a finite gradient."""
    torch.manual_seed(0)
    m = build_model("transformer", {"d_model": 64, "nhead": 4, "num_layers": 2}).to(DEV)
    obs = torch.randn(4, 50, 9, 9, device=DEV)
    m.eval()
    with torch.no_grad():
        pe, _ = m(obs)
    m.train()
    p1, v1 = m(obs)
    p2, _ = m(obs)
    assert not torch.equal(p1, p2) and not torch.equal(p1.detach(), pe)
    assert float((p1.detach() - pe).abs().max()) < 0.6 * float(pe.abs().max()) + 0.5
    (p1.sum() + v1.sum()).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.parameters())
    with pytest.raises(ValueError, match="Expected obs shape"):
        m(torch.zeros(2, 46, 9, 9, device=DEV))
