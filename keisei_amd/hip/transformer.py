"""HIP execution engine of the TransformerModel (BASELINE config 5; reference keisei/training/models/transformer.py:37-95).

Tokens are (B*81, d) row-major, bf16 under autocast or fp32 (parity mode).  The whole network is one autograd node:

  forward : obs -> tokens (padded to 64 planes) | input_proj + row/col embedding | per layer: LayerNorm, in_proj GEMM,
            attention (one wave per board x head, scores and PV on the matrix cores, softmax in registers), out_proj GEMM
            with fused bias + dropout + residual, LayerNorm, FFN GEMMs with fused bias / ReLU / dropout / residual |
            policy GEMM over the flattened 81*d tokens, mean pool + value head
  backward: the same graph reversed; dropout masks are recomputed from (seed, element index), attention probabilities from
            the saved log-sum-exp; weight gradients contract over the token axis (transposed bf16 copies -> the same NT GEMM
            with split-K slabs, reduced in a fixed order).

bf16 mode: every linear layer runs on ``ka_tf_gemm_nt`` (v_mfma_f32_16x16x32_bf16) against bf16 weight copies (plain and
transposed; derived caches, refreshed after an optimiser step).  fp32 mode: the exact-f32 MFMA GEMM ``ka_gemm``.
There is no torch arithmetic on this path beyond the (B,)-sized loss scalars of the caller.
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch
from torch import nn

from keisei_amd import _lib

_call = _lib.call


def _r32(n: int) -> int:
    return (n + 31) // 32 * 32


class _Saved:
    __slots__ = ("B", "T", "train", "seed", "tok", "layers", "x_out", "pooled", "v1", "v", "p_drop")


class TransformerEngine:
    def __init__(self, model: nn.Module) -> None:
        self.model = model
        self._w16: Dict[str, tuple] = {}
        self._w16_key = None
        self._w16_table = None          # (weight pointers, device job table, jobs, tiles) of ka_tf_weights16_multi
        self.weights_epoch = 0
        self._scratch: Dict[str, torch.Tensor] = {}

    def notify_weights_updated(self) -> None:
        self.weights_epoch += 1

    # ------------------------------------------------------------------ parameter views
    def _linears(self):
        m = self.model
        yield "input_proj", m.input_proj.weight, m.input_proj.bias
        for i, lyr in enumerate(m.encoder.layers):
            p = f"encoder.layers.{i}."
            yield p + "self_attn.in_proj", lyr.self_attn.in_proj_weight, lyr.self_attn.in_proj_bias
            yield p + "self_attn.out_proj", lyr.self_attn.out_proj.weight, lyr.self_attn.out_proj.bias
            yield p + "linear1", lyr.linear1.weight, lyr.linear1.bias
            yield p + "linear2", lyr.linear2.weight, lyr.linear2.bias
        yield "policy_fc", m.policy_fc.weight, m.policy_fc.bias

    def _weights16(self, device):
        """bf16 weight copies for the bf16 mode: name -> (W [N][K32], W^T [K][N32]), refreshed when the weights changed."""
        lins = list(self._linears())
        key = (self.weights_epoch, sum(w._version for _, w, _ in lins), lins[0][1].data_ptr(), str(device))
        if key == self._w16_key:
            return self._w16
        st = _lib.stream_ptr(device)
        fresh = False
        for name, w, _ in lins:
            N, K = w.shape
            ent = self._w16.get(name)
            if ent is None or ent[0].device != device:
                ent = (torch.empty(N, _r32(K), dtype=torch.bfloat16, device=device),
                       torch.empty(K, _r32(N), dtype=torch.bfloat16, device=device))
                self._w16[name] = ent
                fresh = True
        multi = (os.environ.get("KA_TF_W16_MULTI", "1") != "0"
                 and all(w.is_contiguous() and w.dtype == torch.float32 for _, w, _ in lins))
        if multi:
            # every layer's two copies in ONE launch, from one read of the weight (ka_tf_weights16_multi)
            ptrs = tuple(w.data_ptr() for _, w, _ in lins)
            tab = self._w16_table
            if tab is None or fresh or tab[0] != ptrs or tab[1].device != device:
                rows, first = [], 0
                for name, w, _ in lins:
                    N, K = w.shape
                    o, oT = self._w16[name]
                    rows.append([w.data_ptr(), o.data_ptr(), oT.data_ptr(), N, K, o.shape[1], oT.shape[1], first])
                    first += ((oT.shape[1] + 63) // 64) * ((o.shape[1] + 63) // 64)
                tab = self._w16_table = (ptrs, torch.tensor(rows, dtype=torch.int64).to(device), len(rows), first)
            _call("ka_tf_weights16_multi", tab[1], tab[2], tab[3], st)
        else:
            for name, w, _ in lins:
                N, K = w.shape
                ent = self._w16[name]
                _call("ka_tf_cast_pad", w, ent[0], N, K, K, _r32(K), _lib.DTYPE_F32, st)
                _call("ka_tf_transpose_pad", w, ent[1], N, K, K, _r32(N), _lib.DTYPE_F32, st)
        self._w16_key = key
        return self._w16

    def _buf(self, tag: str, n: int, dtype, device) -> torch.Tensor:
        t = self._scratch.get(tag)
        if t is None or t.numel() < n or t.dtype != dtype or t.device != device:
            t = torch.empty(n, dtype=dtype, device=device)
            self._scratch[tag] = t
        return t[:n]

    # ------------------------------------------------------------------ linear layers
    def _lin(self, x, name, w, b, T, st, relu=0, drop_p=0.0, seed=0, residual=None, out_f32=False, K=None):
        """act(x W^T + b) [dropout] [+ residual]; x (M, K) of dtype T, result T (or fp32 when out_f32)."""
        M = x.shape[0]
        N, Kw = w.shape
        K = K or Kw
        dev = x.device
        if T == torch.bfloat16:
            w16 = self._weights16(dev)[name][0]
            out = torch.empty(M, N, dtype=torch.float32 if out_f32 else T, device=dev)
            _call("ka_tf_gemm_nt", x, w16, out, b, residual, M, N, w16.shape[1], x.shape[1], w16.shape[1], N,
                  0 if out_f32 else 1, relu, 1, float(drop_p), int(seed), st)
            return out
        out = torch.empty(M, N, device=dev)
        _call("ka_gemm", x, w, out, b, M, N, Kw, x.shape[1], Kw, N, 0, 1, 0, 0, 0, relu, 0, 1, st)
        if drop_p > 0 or residual is not None:
            _call("ka_tf_drop_apply", out, None, residual, out, out.numel(), float(drop_p), int(seed), _lib.DTYPE_F32, st)
        return out

    def _lin_bwd(self, dy, x, name, w, grads, wname, bname, T, st, need_dx=True, dy_is_f32=False, dx_mask=None):
        """dx = dy W, dW = dy^T x, db = colsum(dy) for y = x W^T + b.  dy (M, N) and x (M, Kx >= K) of dtype T (dy fp32 when
        dy_is_f32: the policy logits' gradient)."""
        M, N = dy.shape
        Nw, K = w.shape
        dev = dy.device
        code = _lib.dtype_code(T)
        dy_code = _lib.DTYPE_F32 if dy_is_f32 else code
        db = torch.empty(N, device=dev)
        grads[bname] = db
        dW = torch.empty(N, K, device=dev)
        dx = None
        tn_ok = (T == torch.bfloat16 and not dy_is_f32 and N % 8 == 0 and K % 8 == 0 and x.shape[1] % 8 == 0
                 and os.environ.get("KA_TF_TN", "1") != "0")
        if tn_ok:
            # weight gradient straight from the row-major activations (LDS transpose reads: no transposed copies), and the
            # bias gradient in the same launch (column sums of dy against an all-ones fragment: no second pass over dy)
            tiles = ((N + 127) // 128) * ((K + 127) // 128)
            want = max(1, min(M // 64, (512 + tiles - 1) // tiles))
            ns = _lib.query("ka_tf_gemm_tn_slabs", M, want)
            if ns == 1:
                _call("ka_tf_gemm_tn_bias", dy, x, dW, db, M, N, K, N, x.shape[1], K, 1, st)
            else:
                slab = self._buf("slab", ns * N * K, torch.float32, dev)
                cs = self._buf("colsum", ns * N, torch.float32, dev)
                _call("ka_tf_gemm_tn_bias", dy, x, slab, cs, M, N, K, N, x.shape[1], K, want, st)
                _call("ka_reduce_slabs2", slab, dW, N * K, cs, db, N, ns, st)       # weight- and bias-gradient partials in one launch
        else:
            nsb = max(1, min(1024, (M + 255) // 256))
            part = self._buf("colsum", nsb * N, torch.float32, dev)
            _call("ka_tf_colsum", dy, part, db, M, N, nsb, dy_code, st)
        if T == torch.bfloat16 and not tn_ok:
            Mp = _r32(M)
            dyT = self._buf("dyT", N * Mp, torch.bfloat16, dev)
            xT = self._buf("xT", x.shape[1] * Mp, torch.bfloat16, dev)
            _call("ka_tf_transpose_pad", dy, dyT, M, N, N, Mp, dy_code, st)
            _call("ka_tf_transpose_pad", x, xT, M, x.shape[1], x.shape[1], Mp, code, st)
            tiles = ((N + 127) // 128) * ((K + 127) // 128)
            want = max(1, min(Mp // 32, (512 + tiles - 1) // tiles))
            ns = _lib.query("ka_tf_gemm_nt_slabs", Mp, want)
            if ns == 1:
                _call("ka_tf_gemm_nt", dyT, xT, dW, None, None, N, K, Mp, Mp, Mp, K, 0, 0, 1, 0.0, 0, st)
            else:
                slab = self._buf("slab", ns * N * K, torch.float32, dev)
                _call("ka_tf_gemm_nt", dyT, xT, slab, None, None, N, K, Mp, Mp, Mp, K, 0, 0, want, 0.0, 0, st)
                _call("ka_reduce_slabs", slab, dW, ns, N * K, 0, st)
        if T == torch.bfloat16:
            if need_dx:
                wT16 = self._weights16(dev)[name][1]                  # [K][N32]
                if dy_is_f32 or N % 32:
                    d16 = self._buf("dy16", M * _r32(N), torch.bfloat16, dev)
                    _call("ka_tf_cast_pad", dy, d16, M, N, N, _r32(N), dy_code, st)
                    a, lda = d16, _r32(N)
                else:
                    a, lda = dy, N
                dx = torch.empty(M, K, dtype=T, device=dev)
                if dx_mask is not None:          # (act, p, seed): the ReLU + dropout that precede this layer, in the GEMM epilogue
                    act, pm, sm = dx_mask
                    _call("ka_tf_gemm_nt_masked", a, wT16, dx, act, M, K, _r32(N), lda, wT16.shape[1], K, float(pm), int(sm), st)
                else:
                    _call("ka_tf_gemm_nt", a, wT16, dx, None, None, M, K, _r32(N), lda, wT16.shape[1], K, 1, 0, 1, 0.0, 0, st)
        else:
            ns = max(1, min(256, (M + 511) // 512))
            if ns == 1:
                _call("ka_gemm", dy, x, dW, None, N, K, M, N, x.shape[1], K, 1, 0, 0, 0, 0, 0, 0, 1, st)
            else:
                slab = self._buf("slab", ns * N * K, torch.float32, dev)
                _call("ka_gemm", dy, x, slab, None, N, K, M, N, x.shape[1], K, 1, 0, 0, 0, 0, 0, 0, ns, st)
                _call("ka_reduce_slabs", slab, dW, ns, N * K, 0, st)
            if need_dx:
                dx = torch.empty(M, K, device=dev)
                _call("ka_gemm", dy, w, dx, None, M, K, N, N, K, K, 0, 0, 0, 0, 0, 0, 0, 1, st)
        grads[wname] = dW
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, obs: torch.Tensor, train: bool, keep: bool, T: torch.dtype):
        m = self.model
        dev = obs.device
        st = _lib.stream_ptr(dev)
        code = _lib.dtype_code(T)
        B = obs.shape[0]
        M = B * 81
        d = m.input_proj.out_features
        if obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.float().contiguous()
        seed = int(torch.empty((), dtype=torch.int64).random_()) & 0x3FFFFFFFFFFFFFFF if train else 0
        sv = _Saved()
        sv.B, sv.T, sv.train, sv.seed = B, T, train, seed
        # tokens: (B, 81, 64) = NCHW -> NHWC with the 50 planes zero-padded to the MFMA K step
        tok = torch.empty(M, 64, dtype=T, device=dev)
        _call("ka_obs_to_nhwc", obs, None, tok, B, m.OBS_CHANNELS, 64, code, st)
        x = self._lin(tok, "input_proj", m.input_proj.weight, m.input_proj.bias, T, st)
        _call("ka_tf_add_pos", x, m.row_embed.weight, m.col_embed.weight, B, d, code, st)
        sv.tok, sv.layers = tok, []
        for i, lyr in enumerate(m.encoder.layers):
            pre = f"encoder.layers.{i}."
            H = lyr.self_attn.num_heads
            p_attn = float(lyr.self_attn.dropout) if train else 0.0
            p1, p2, pf = (float(lyr.dropout1.p), float(lyr.dropout2.p), float(lyr.dropout.p)) if train else (0.0, 0.0, 0.0)
            s_base = seed + 7919 * (i + 1)
            mu1 = torch.empty(M, device=dev); rs1 = torch.empty(M, device=dev)
            h1 = torch.empty(M, d, dtype=T, device=dev)
            _call("ka_tf_layernorm_fwd", x, lyr.norm1.weight, lyr.norm1.bias, h1, mu1, rs1, M, d, float(lyr.norm1.eps), code, st)
            qkv = self._lin(h1, pre + "self_attn.in_proj", lyr.self_attn.in_proj_weight, lyr.self_attn.in_proj_bias, T, st)
            attn = torch.empty(M, d, dtype=T, device=dev)
            lse = torch.empty(B, H, 81, device=dev)
            _call("ka_tf_attention_fwd", qkv, attn, lse, B, H, d // H, p_attn, s_base + 1, code, st)
            x_mid = self._lin(attn, pre + "self_attn.out_proj", lyr.self_attn.out_proj.weight, lyr.self_attn.out_proj.bias, T, st,
                              drop_p=p1, seed=s_base + 2, residual=x)
            mu2 = torch.empty(M, device=dev); rs2 = torch.empty(M, device=dev)
            h2 = torch.empty(M, d, dtype=T, device=dev)
            _call("ka_tf_layernorm_fwd", x_mid, lyr.norm2.weight, lyr.norm2.bias, h2, mu2, rs2, M, d, float(lyr.norm2.eps), code, st)
            f = self._lin(h2, pre + "linear1", lyr.linear1.weight, lyr.linear1.bias, T, st, relu=1, drop_p=pf, seed=s_base + 3)
            x_out = self._lin(f, pre + "linear2", lyr.linear2.weight, lyr.linear2.bias, T, st, drop_p=p2, seed=s_base + 4,
                              residual=x_mid)
            if keep:
                sv.layers.append((x, h1, mu1, rs1, qkv, attn, lse, x_mid, h2, mu2, rs2, f, (p_attn, p1, p2, pf), s_base))
            x = x_out
        if m.encoder.norm is not None:
            raise _lib.KeiseiHipError("TransformerEncoder with a final norm is not what the reference builds")
        policy = self._lin(x.view(B, 81 * d), "policy_fc", m.policy_fc.weight, m.policy_fc.bias, T, st, out_f32=True)
        pooled = torch.empty(B, d, device=dev)
        _call("ka_tf_mean_pool", x, pooled, B, d, code, st)
        v1 = torch.empty(B, d, device=dev)
        w1, w2 = m.value_fc1.weight, m.value_fc2.weight
        _call("ka_gemm", pooled, w1, v1, m.value_fc1.bias, B, d, d, d, d, d, 0, 1, 0, 0, 0, 1, 0, 1, st)
        v = torch.empty(B, 1, device=dev)
        _call("ka_gemm", v1, w2, v, m.value_fc2.bias, B, 1, d, d, d, 1, 0, 1, 0, 0, 0, 0, 0, 1, st)
        _call("ka_tf_tanh", v, B, st)
        if keep:
            sv.x_out, sv.pooled, sv.v1, sv.v = x, pooled, v1, v
        return policy, v, (sv if keep else None)

    # ------------------------------------------------------------------ backward
    def _small_lin_bwd(self, dy, x, lin, grads, name, st, need_dx=True):
        """fp32 head layers (B rows): dW = dy^T x, db = colsum(dy), dx = dy W on the exact-f32 GEMM."""
        M, N = dy.shape
        K = lin.weight.shape[1]
        dev = dy.device
        dW = torch.empty(N, K, device=dev)
        _call("ka_gemm", dy, x, dW, None, N, K, M, N, K, K, 1, 0, 0, 0, 0, 0, 0, 1, st)
        db = torch.empty(N, device=dev)
        part = self._buf("colsum", N, torch.float32, dev)
        _call("ka_tf_colsum", dy, part, db, M, N, 1, _lib.DTYPE_F32, st)
        grads[name + ".weight"], grads[name + ".bias"] = dW, db
        if not need_dx:
            return None
        dx = torch.empty(M, K, device=dev)
        _call("ka_gemm", dy, lin.weight, dx, None, M, K, N, N, K, K, 0, 0, 0, 0, 0, 0, 0, 1, st)
        return dx

    def backward(self, sv: _Saved, dpolicy: Optional[torch.Tensor], dvalue: Optional[torch.Tensor]) -> Dict[str, torch.Tensor]:
        m = self.model
        B, T = sv.B, sv.T
        M = B * 81
        d = m.input_proj.out_features
        dev = sv.tok.device
        st = _lib.stream_ptr(dev)
        code = _lib.dtype_code(T)
        grads: Dict[str, torch.Tensor] = {}
        # ---- value head
        dpooled = None
        if dvalue is not None:
            dz2 = torch.empty(B, 1, device=dev)
            _call("ka_tf_tanh_bwd", dvalue.float().contiguous(), sv.v, dz2, B, st)
            dv1 = self._small_lin_bwd(dz2, sv.v1, m.value_fc2, grads, "value_fc2", st)
            _call("ka_relu_mask", dv1, sv.v1, dv1.numel(), st)
            dpooled = self._small_lin_bwd(dv1, sv.pooled, m.value_fc1, grads, "value_fc1", st)
        else:
            for n, p in (("value_fc1", m.value_fc1), ("value_fc2", m.value_fc2)):
                grads[n + ".weight"], grads[n + ".bias"] = torch.zeros_like(p.weight), torch.zeros_like(p.bias)
        # ---- policy head
        dflat = None
        if dpolicy is not None:
            dpol = dpolicy.float().contiguous()
            dflat = self._lin_bwd(dpol, sv.x_out.view(B, 81 * d), "policy_fc", m.policy_fc.weight, grads,
                                  "policy_fc.weight", "policy_fc.bias", T, st, dy_is_f32=True)
        else:
            grads["policy_fc.weight"], grads["policy_fc.bias"] = torch.zeros_like(m.policy_fc.weight), torch.zeros_like(m.policy_fc.bias)
        dx = torch.empty(M, d, dtype=T, device=dev)
        _call("ka_tf_head_grad", dpooled, dflat, dx, B, d, code, st)
        # ---- encoder layers, last first
        nparts = _lib.query("ka_tf_layernorm_parts", M)
        lnws = self._buf("lnws", (nparts + 1) * 2 * d, torch.float32, dev)
        g2_pre = None
        for i in range(len(sv.layers) - 1, -1, -1):
            lyr = m.encoder.layers[i]
            pre = f"encoder.layers.{i}."
            (x_in, h1, mu1, rs1, qkv, attn, lse, x_mid, h2, mu2, rs2, f, (p_attn, p1, p2, pf), s_base) = sv.layers[i]
            H = lyr.self_attn.num_heads
            g2 = dx
            if p2 > 0:
                if g2_pre is not None:           # written by the LayerNorm backward of the layer above, in the same pass as dx
                    g2 = g2_pre
                else:
                    g2 = torch.empty_like(dx)
                    _call("ka_tf_drop_apply", dx, None, None, g2, dx.numel(), p2, s_base + 4, code, st)
            fused_mask = T == torch.bfloat16
            df = self._lin_bwd(g2, f, pre + "linear2", lyr.linear2.weight, grads, pre + "linear2.weight", pre + "linear2.bias", T, st,
                               dx_mask=(f, pf, s_base + 3) if fused_mask else None)
            if not fused_mask:
                _call("ka_tf_drop_apply", df, f, None, df, df.numel(), pf, s_base + 3, code, st)   # through dropout and ReLU
            dh2 = self._lin_bwd(df, h2, pre + "linear1", lyr.linear1.weight, grads, pre + "linear1.weight", pre + "linear1.bias", T, st)
            dxm = torch.empty_like(dx)
            dg, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
            fuse_ln = os.environ.get("KA_TF_LN_DROP", "1") != "0"
            g1 = torch.empty_like(dxm) if (p1 > 0 and fuse_ln) else None   # dxm through the dropout after out_proj, same pass
            _call("ka_tf_layernorm_bwd_drop", dh2, x_mid, lyr.norm2.weight, mu2, rs2, dx, dxm, g1, float(p1), int(s_base + 2),
                  lnws, dg, db, M, d, code, st)
            grads[pre + "norm2.weight"], grads[pre + "norm2.bias"] = dg, db
            if g1 is None:
                g1 = dxm
                if p1 > 0:
                    g1 = torch.empty_like(dxm)
                    _call("ka_tf_drop_apply", dxm, None, None, g1, dxm.numel(), p1, s_base + 2, code, st)
            dattn = self._lin_bwd(g1, attn, pre + "self_attn.out_proj", lyr.self_attn.out_proj.weight, grads,
                                  pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias", T, st)
            dqkv = torch.empty_like(qkv)
            # (the forward's output `attn` gives D = rowsum(dattn * attn): dQ and dK / dV run as two independent launches)
            _call("ka_tf_attention_bwd_o", qkv, attn, dattn, lse, dqkv, B, H, d // H, p_attn, s_base + 1, code, st)
            dh1 = self._lin_bwd(dqkv, h1, pre + "self_attn.in_proj", lyr.self_attn.in_proj_weight, grads,
                                pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias", T, st)
            dx_new = torch.empty_like(dx)
            dg, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
            g2_pre = None
            p2_below, seed_below = 0.0, 0
            if i > 0:
                p2_below, seed_below = float(sv.layers[i - 1][12][2]), int(sv.layers[i - 1][13]) + 4
                if p2_below > 0 and fuse_ln:
                    g2_pre = torch.empty_like(dx)    # the layer below receives dx through the dropout after its linear2
            _call("ka_tf_layernorm_bwd_drop", dh1, x_in, lyr.norm1.weight, mu1, rs1, dxm, dx_new, g2_pre, p2_below, seed_below,
                  lnws, dg, db, M, d, code, st)
            grads[pre + "norm1.weight"], grads[pre + "norm1.bias"] = dg, db
            dx = dx_new
        # ---- embeddings and input projection
        drow, dcol = torch.empty(9, d, device=dev), torch.empty(9, d, device=dev)
        _call("ka_tf_pos_grad", dx, self._buf("pos", 65 * 81 * d, torch.float32, dev), drow, dcol, B, d, code, st)
        grads["row_embed.weight"], grads["col_embed.weight"] = drow, dcol
        tmp: Dict[str, torch.Tensor] = {}
        w_in = m.input_proj.weight
        if T == torch.bfloat16:
            # contraction over tokens against the 64-plane padded tokens: dW (d, 64), the reference's 50 columns are a slice
            wpad = torch.empty(d, 64, device=dev)           # shape carrier for _lin_bwd (K = 64)
            self._lin_bwd(dx, sv.tok, "input_proj", wpad, tmp, "w", "b", T, st, need_dx=False)
            grads["input_proj.weight"] = tmp["w"][:, :w_in.shape[1]].contiguous()
        else:
            self._lin_bwd(dx, sv.tok, "input_proj", w_in, tmp, "w", "b", T, st, need_dx=False)
            grads["input_proj.weight"] = tmp["w"]
        grads["input_proj.bias"] = tmp["b"]
        return grads


class _TransformerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: TransformerEngine, obs, train, keep, T, names, *params):
        policy, value, saved = engine.forward(obs, train, keep, T)
        ctx.engine, ctx.saved, ctx.names = engine, saved, names
        ctx.set_materialize_grads(False)
        return policy, value

    @staticmethod
    def backward(ctx, dpolicy, dvalue):
        if ctx.saved is None:
            raise RuntimeError("transformer HIP backward called without saved activations")
        with torch.cuda.device(ctx.saved.tok.device):
            grads = ctx.engine.backward(ctx.saved, dpolicy, dvalue)
        ctx.saved = None
        return (None, None, None, None, None, None, *[grads.get(n) for n in ctx.names])


def supported(model: nn.Module) -> Optional[str]:
    """None when the HIP path covers this configuration, else the reason."""
    d = model.input_proj.out_features
    lyr = model.encoder.layers[0]
    H = lyr.self_attn.num_heads
    if d % 32:
        return f"d_model {d} is not a multiple of 32"
    if d // H > 64:
        return f"head dimension {d // H} > 64"
    if not lyr.norm_first or getattr(lyr, "activation_relu_or_gelu", 1) != 1:
        return "only the reference's pre-norm ReLU encoder layer is implemented"
    return None


def run_model(model: nn.Module, obs: torch.Tensor):
    """Forward of TransformerModel on a CUDA/HIP device: (policy_logits (B, 11259) fp32, value (B, 1) fp32)."""
    why = supported(model)
    if why is not None:
        raise _lib.KeiseiHipError(f"TransformerModel HIP path: {why} (there is no fallback for GPU tensors)")
    engine = getattr(model, "_hip_engine", None)
    if engine is None:
        engine = TransformerEngine(model)
        object.__setattr__(model, "_hip_engine", engine)
    T = torch.float32
    if torch.is_autocast_enabled("cuda"):
        if torch.get_autocast_dtype("cuda") != torch.bfloat16:
            raise _lib.KeiseiHipError("the HIP path supports fp32 and bf16 autocast only")
        T = torch.bfloat16
    named = list(model.named_parameters())
    keep = torch.is_grad_enabled() and any(p.requires_grad for _, p in named)
    names = tuple(n for n, _ in named)
    with torch.autocast("cuda", enabled=False), torch.cuda.device(obs.device):
        return _TransformerFunction.apply(engine, obs, model.training, keep, T, names, *[p for _, p in named])
