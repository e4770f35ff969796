"""CPU oracle for the Keisei PPO-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``keisei_amd/`` imports this file.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
use it, and there only as the checker / the timed CPU baseline -- never as the
product path.

This is a from-scratch *functional* restatement (plain fp32 PyTorch CPU ops over a
``state_dict``) of the algorithm in the reference files below.  The arithmetic of the
reference lives in a third-party dependency -- PyTorch (``torch`` pinned 2.11.0 in the
reference's ``uv.lock``; 2.10.0+rocm7.0 CPU backend here) -- so parity is anchored on
the reference's own call sites and on golden vectors produced by importing the
reference in the dev container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
``tests/test_oracle_golden.py`` pins every function here against those vectors and
against the reference tests' hand-computed known answers.

Reference citations (relative to the reference repo root):
  * SE-ResNet forward ......... keisei/training/models/se_resnet.py:40-159
  * global pool ............... keisei/training/models/se_resnet.py:93-98
  * value adapters ............ keisei/training/value_adapter.py:76-126
  * clip loss / CE / metrics .. keisei/training/katago_ppo.py:33-78
  * minibatch loss ............ keisei/training/katago_ppo.py:849-924
  * clip + Adam ............... keisei/training/katago_ppo.py:926-933 (torch.optim.Adam defaults)
  * GAE ....................... keisei/training/gae.py:8-296
  * advantage normalisation ... keisei/training/katago_ppo.py:797-798
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
N_MOVE_TYPES = 139
N_ACTIONS = 81 * N_MOVE_TYPES


# --------------------------------------------------------------------------- model

@dataclass(frozen=True)
class NetShape:
    """Mirror of SEResNetParams (se_resnet.py:15-37); plain ints only."""
    num_blocks: int = 40
    channels: int = 256
    se_reduction: int = 16
    global_pool_channels: int = 128
    policy_channels: int = 32
    value_fc_size: int = 256
    score_fc_size: int = 128
    obs_channels: int = 50


def state_dict_spec(s: NetShape) -> Dict[str, tuple]:
    """Key -> shape contract of the reference model's state_dict."""
    C, G, R = s.channels, s.global_pool_channels, s.channels // s.se_reduction
    spec: Dict[str, tuple] = {}

    def bn(prefix: str, n: int) -> None:
        spec[prefix + ".weight"] = (n,)
        spec[prefix + ".bias"] = (n,)
        spec[prefix + ".running_mean"] = (n,)
        spec[prefix + ".running_var"] = (n,)
        spec[prefix + ".num_batches_tracked"] = ()

    spec["input_conv.weight"] = (C, s.obs_channels, 3, 3)
    bn("input_bn", C)
    for i in range(s.num_blocks):
        p = f"blocks.{i}."
        spec[p + "conv1.weight"] = (C, C, 3, 3)
        bn(p + "bn1", C)
        spec[p + "conv2.weight"] = (C, C, 3, 3)
        bn(p + "bn2", C)
        spec[p + "global_fc.0.weight"] = (G, 3 * C)
        spec[p + "global_fc.0.bias"] = (G,)
        spec[p + "global_fc.2.weight"] = (C, G)
        spec[p + "global_fc.2.bias"] = (C,)
        spec[p + "se_fc1.weight"] = (R, C)
        spec[p + "se_fc1.bias"] = (R,)
        spec[p + "se_fc2.weight"] = (2 * C, R)
        spec[p + "se_fc2.bias"] = (2 * C,)
    spec["policy_conv1.weight"] = (s.policy_channels, C, 1, 1)
    bn("policy_bn1", s.policy_channels)
    spec["policy_conv2.weight"] = (N_MOVE_TYPES, s.policy_channels, 1, 1)
    spec["policy_conv2.bias"] = (N_MOVE_TYPES,)
    spec["value_fc1.weight"] = (s.value_fc_size, 3 * C)
    spec["value_fc1.bias"] = (s.value_fc_size,)
    spec["value_fc2.weight"] = (3, s.value_fc_size)
    spec["value_fc2.bias"] = (3,)
    spec["score_fc1.weight"] = (s.score_fc_size, 3 * C)
    spec["score_fc1.bias"] = (s.score_fc_size,)
    spec["score_fc2.weight"] = (1, s.score_fc_size)
    spec["score_fc2.bias"] = (1,)
    return spec


def synth_state_dict(s: NetShape, salt: int = 0) -> Dict[str, torch.Tensor]:
    """Closed-form deterministic weights (no 214 MB fixtures).

    value(key, i) = amp(key) * sin(0.7311 * i + phase(key)), with amp chosen so
    activations stay O(1) through 40 blocks; BN gamma near 1, running_var > 0.
    Both the oracle and the HIP path are filled from this generator in tests.
    """
    out: Dict[str, torch.Tensor] = {}
    for k_idx, (key, shape) in enumerate(state_dict_spec(s).items()):
        n = int(np.prod(shape)) if shape else 1
        i = torch.arange(n, dtype=torch.float64)
        phase = 0.37 * (k_idx + 1) + 0.011 * salt
        wave = torch.sin(0.7311 * i + phase) + 0.35 * torch.sin(0.1234567 * i * (1 + (k_idx % 7)) + 2.0 * phase)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        if key.endswith("running_var"):
            v = 1.0 + 0.25 * wave
        elif key.endswith("running_mean"):
            v = 0.1 * wave
        elif ".bn" in key or key.startswith(("input_bn", "policy_bn1")):
            v = (1.0 + 0.1 * wave) if key.endswith("weight") else 0.05 * wave
        elif key.endswith("bias"):
            v = 0.05 * wave
        else:
            fan_in = int(np.prod(shape[1:]))
            v = wave * math.sqrt(0.5 / fan_in)
        out[key] = v.to(torch.float32).reshape(shape)
    return out


def _hash_uniform(n: int, salt: int) -> torch.Tensor:
    """n doubles in [-1, 1): a counter-based integer hash (splitmix64 finaliser) of (salt, index).  Pure integer
    arithmetic modulo 2^64, so every platform reproduces the same values bit for bit."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64((salt * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))       # [0, 1)
    return torch.from_numpy(2.0 * u - 1.0)


def init_like_state_dict(s: NetShape, salt: int = 0) -> Dict[str, torch.Tensor]:
    """Closed-form weights with the *statistics* of the reference's default initialisation (PyTorch
    defaults, kaiming_uniform(a=sqrt 5) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv / linear weights and biases),
    drawn from a counter-based hash instead of an RNG so that both sides of a parity test rebuild the 214 MB of the
    40x256 model from (shape, salt) alone.  BatchNorm gets a non-trivial affine (gamma 1 +- 0.1, beta +- 0.05) and
    running statistics (mean +- 0.1, var 1 +- 0.25).

    Unlike ``synth_state_dict`` (sums of two sines of the flat index: every weight matrix is numerically rank 4, and a
    40-block tower of them amplifies fp32 rounding into 50 % gradient differences between fp32 and fp64 runs of the
    reference itself), these weights are full rank and the network is as well conditioned as a freshly initialised one."""
    out: Dict[str, torch.Tensor] = {}
    for k_idx, (key, shape) in enumerate(state_dict_spec(s).items()):
        n = int(np.prod(shape)) if shape else 1
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        u = _hash_uniform(n, 1000 * salt + k_idx + 1)
        if key.endswith("running_var"):
            v = 1.0 + 0.25 * u
        elif key.endswith("running_mean"):
            v = 0.1 * u
        elif ".bn" in key or key.startswith(("input_bn", "policy_bn1")):
            v = (1.0 + 0.1 * u) if key.endswith("weight") else 0.05 * u
        elif key.endswith("bias"):
            w_shape = state_dict_spec(s)[key[:-4] + "weight"]
            v = u / math.sqrt(int(np.prod(w_shape[1:])))
        else:
            v = u / math.sqrt(int(np.prod(shape[1:])))
        out[key] = v.to(torch.float32).reshape(shape)
    return out


def closed_form_cotangents(batch: int) -> tuple:
    """Deterministic output cotangents (policy (B,9,9,139), value (B,3), score (B,1)) for gradient fixtures: nothing to
    store, both sides rebuild them."""
    cp = _hash_uniform(batch * N_ACTIONS, 7001).to(torch.float32).reshape(batch, 9, 9, N_MOVE_TYPES)
    cv = _hash_uniform(batch * 3, 7002).to(torch.float32).reshape(batch, 3)
    cs = _hash_uniform(batch, 7003).to(torch.float32).reshape(batch, 1)
    return cp, cv, cs


def hash_fill(sd: Dict[str, torch.Tensor], salt: int = 0) -> Dict[str, torch.Tensor]:
    """Init-like closed-form weights for ANY state_dict (the scalar-contract models): matrices U(-1, 1) / sqrt(fan_in),
    1-D tensors named like norm scales 1 +- 0.1, other 1-D tensors +- 0.05 (embeddings: +- 0.5), from the counter-based
    hash -- full rank, reproducible everywhere, nothing to store."""
    out: Dict[str, torch.Tensor] = {}
    for k_idx, (key, t) in enumerate(sd.items()):
        if not t.dtype.is_floating_point:
            out[key] = t.clone()
            continue
        u = _hash_uniform(t.numel(), 5000 + 1000 * salt + k_idx)
        if "embed" in key:
            v = 0.5 * u
        elif t.ndim >= 2:
            v = u / math.sqrt(max(1, t[0].numel()))
        elif "norm" in key and key.endswith("weight"):
            v = 1.0 + 0.1 * u
        else:
            v = 0.05 * u
        out[key] = v.to(t.dtype).reshape(t.shape)
    return out


def closed_form_fill(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Generic closed-form weights for any state_dict (used for the scalar-contract models,
    whose policy_fc is too large to commit): float tensors get amp*sin(.), norm scales ~1,
    variances > 0, integer buffers are kept."""
    out: Dict[str, torch.Tensor] = {}
    for k_idx, (key, t) in enumerate(sd.items()):
        if not t.dtype.is_floating_point:
            out[key] = t.clone()
            continue
        n = t.numel()
        i = torch.arange(n, dtype=torch.float64)
        wave = torch.sin(0.7311 * i + 0.37 * (k_idx + 1)) + 0.35 * torch.sin(0.0917 * i * (1 + k_idx % 5))
        is_norm = any(s in key for s in ("norm", "bn", "trunk.1", "trunk.4", "trunk.7"))
        if key.endswith("running_var"):
            v = 1.0 + 0.25 * wave
        elif key.endswith("running_mean"):
            v = 0.1 * wave
        elif is_norm and key.endswith("weight") and t.ndim == 1:
            v = 1.0 + 0.1 * wave
        elif t.ndim <= 1:
            v = 0.05 * wave
        else:
            v = wave * math.sqrt(1.0 / max(1, t[0].numel()))
        out[key] = v.to(t.dtype).reshape(t.shape)
    return out


def global_pool(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> (B,3C) = [mean | max | population std].  se_resnet.py:93-98."""
    flat = x.flatten(2)
    mu = flat.mean(dim=2)
    mx = flat.amax(dim=2)
    sd = flat.std(dim=2, correction=0)
    return torch.cat((mu, mx, sd), dim=1)


def _bn(x, sd, prefix, train, momentum, update_running):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train and not update_running:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, w, b, training=train, momentum=momentum, eps=BN_EPS)
    if train and update_running:
        sd[prefix + ".num_batches_tracked"] += 1
    return y


def block_forward(sd, prefix: str, x: torch.Tensor, train: bool,
                  momentum: float = 0.1, update_running: bool = False) -> torch.Tensor:
    """One GlobalPoolBiasBlock (se_resnet.py:68-90), functional."""
    C = x.shape[1]
    h = F.conv2d(x, sd[prefix + "conv1.weight"], padding=1)
    h = torch.relu(_bn(h, sd, prefix + "bn1", train, momentum, update_running))
    g = global_pool(x)
    g = torch.relu(F.linear(g, sd[prefix + "global_fc.0.weight"], sd[prefix + "global_fc.0.bias"]))
    g = F.linear(g, sd[prefix + "global_fc.2.weight"], sd[prefix + "global_fc.2.bias"])
    h = h + g[:, :, None, None]
    z = _bn(F.conv2d(h, sd[prefix + "conv2.weight"], padding=1), sd, prefix + "bn2",
            train, momentum, update_running)
    sq = z.flatten(2).mean(dim=2)
    e = torch.relu(F.linear(sq, sd[prefix + "se_fc1.weight"], sd[prefix + "se_fc1.bias"]))
    e = F.linear(e, sd[prefix + "se_fc2.weight"], sd[prefix + "se_fc2.bias"])
    gate, shift = e[:, :C], e[:, C:]
    u = z * torch.sigmoid(gate)[:, :, None, None] + shift[:, :, None, None]
    return torch.relu(u + x)


def seresnet_forward(sd, obs: torch.Tensor, num_blocks: int, train: bool,
                     momentum: float = 0.1, update_running: bool = False):
    """SEResNetModel._forward_impl (se_resnet.py:132-159).

    Returns (policy_logits (B,9,9,139), value_logits (B,3), score_lead (B,1)).
    """
    c_in = sd["input_conv.weight"].shape[1]
    if obs.ndim != 4 or tuple(obs.shape[1:]) != (c_in, 9, 9):
        raise ValueError(f"Expected obs shape (batch, {c_in}, 9, 9), got {tuple(obs.shape)}")
    x = F.conv2d(obs, sd["input_conv.weight"], padding=1)
    x = torch.relu(_bn(x, sd, "input_bn", train, momentum, update_running))
    for i in range(num_blocks):
        x = block_forward(sd, f"blocks.{i}.", x, train, momentum, update_running)
    p = F.conv2d(x, sd["policy_conv1.weight"])
    p = torch.relu(_bn(p, sd, "policy_bn1", train, momentum, update_running))
    p = F.conv2d(p, sd["policy_conv2.weight"], sd["policy_conv2.bias"])
    policy = p.permute(0, 2, 3, 1)
    pool = global_pool(x)
    v = torch.relu(F.linear(pool, sd["value_fc1.weight"], sd["value_fc1.bias"]))
    v = F.linear(v, sd["value_fc2.weight"], sd["value_fc2.bias"])
    s = torch.relu(F.linear(pool, sd["score_fc1.weight"], sd["score_fc1.bias"]))
    s = F.linear(s, sd["score_fc2.weight"], sd["score_fc2.bias"])
    return policy, v, s


def relu_margin(sd, obs: torch.Tensor, num_blocks: int) -> float:
    """Smallest non-zero |ReLU input| of a train-mode fp64 forward: how far the batch is from a ReLU knife edge.  Two
    fp32 implementations agree on every ReLU mask (and hence on gradients to ~1e-6) when this is well above their
    rounding noise (~1e-6 for O(1) activations); fixtures and smoke() pick their batch by it."""
    seen = {"min": float("inf")}
    real = torch.relu

    def rec(x):
        a = x.detach().abs()
        a = a[a > 0]
        if a.numel():
            seen["min"] = min(seen["min"], float(a.min()))
        return real(x)

    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    torch.relu = rec
    try:
        with torch.no_grad():
            seresnet_forward(sd64, obs.double(), num_blocks, train=True, momentum=0.0)
    finally:
        torch.relu = real
    return seen["min"]


def seresnet_policy_bf16_storage(sd, obs: torch.Tensor, num_blocks: int, train: bool) -> torch.Tensor:
    """fp32 math with bf16 rounding at exactly the points where the HIP bf16 mode stores bf16
    (conv operands, y1/y2/out activations, the fused relu(bn1)+g conv2 input); BN statistics come
    from the unrounded conv results, as in the conv epilogue.  This DEFINES the bf16 mode's stated
    numerics: tests hold the HIP bf16 path to a few bf16 ulps of this emulation, and report the
    (inherent, input-dependent) distance of both from the fp32 reference side by side."""
    def q(t):
        return t.bfloat16().float()

    def coeffs(y, pre):
        if train:
            mu, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
        else:
            mu, var = sd[pre + ".running_mean"], sd[pre + ".running_var"]
        sc = sd[pre + ".weight"] / torch.sqrt(var + BN_EPS)
        return sc, sd[pre + ".bias"] - mu * sc

    def aff(y, sc, sh):
        return y * sc[None, :, None, None] + sh[None, :, None, None]

    y0 = F.conv2d(q(obs), q(sd["input_conv.weight"]), padding=1)
    sc, sh = coeffs(y0, "input_bn")
    x = q(torch.relu(aff(q(y0), sc, sh)))
    for i in range(num_blocks):
        pre = f"blocks.{i}."
        C = x.shape[1]
        y1 = F.conv2d(x, q(sd[pre + "conv1.weight"]), padding=1)
        sc1, sh1 = coeffs(y1, pre + "bn1")
        g = torch.relu(F.linear(global_pool(x), sd[pre + "global_fc.0.weight"], sd[pre + "global_fc.0.bias"]))
        g = F.linear(g, sd[pre + "global_fc.2.weight"], sd[pre + "global_fc.2.bias"])
        h = q(torch.relu(aff(q(y1), sc1, sh1)) + g[:, :, None, None])
        y2 = F.conv2d(h, q(sd[pre + "conv2.weight"]), padding=1)
        sc2, sh2 = coeffs(y2, pre + "bn2")
        sqz = sc2 * y2.mean(dim=(2, 3)) + sh2
        e = torch.relu(F.linear(sqz, sd[pre + "se_fc1.weight"], sd[pre + "se_fc1.bias"]))
        e = F.linear(e, sd[pre + "se_fc2.weight"], sd[pre + "se_fc2.bias"])
        x = q(torch.relu(aff(q(y2), sc2, sh2) * torch.sigmoid(e[:, :C])[:, :, None, None] + e[:, C:, None, None] + x))
    p1 = F.conv2d(x, sd["policy_conv1.weight"])
    scp, shp = coeffs(p1, "policy_bn1")
    p = F.conv2d(torch.relu(aff(p1, scp, shp)), sd["policy_conv2.weight"], sd["policy_conv2.bias"])
    return p.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------- losses

def scalar_value(value_logits: torch.Tensor) -> torch.Tensor:
    """P(W) - P(L).  katago_ppo.py:533-541, value_adapter.py:76-79."""
    p = torch.softmax(value_logits, dim=-1)
    return p[:, 0] - p[:, 2]


def scalar_value_blended(value_logits, score_lead, alpha: float) -> torch.Tensor:
    """value_adapter.py:81-96."""
    w = scalar_value(value_logits)
    if alpha == 0.0:
        return w
    return (1 - alpha) * w + alpha * score_lead.squeeze(-1).clamp(-1, 1)


def clip_surrogate(new_lp, old_lp, adv, eps: float) -> torch.Tensor:
    """katago_ppo.py:33-43."""
    r = torch.exp(new_lp - old_lp)
    return -torch.minimum(r * adv, r.clamp(1 - eps, 1 + eps) * adv).mean()


def wdl_ce(value_logits, cats) -> torch.Tensor:
    """katago_ppo.py:46-57 / value_adapter.py:113-119 (all-ignored -> graph-connected 0)."""
    if not bool((cats >= 0).any()):
        return value_logits.sum() * 0.0
    return F.cross_entropy(value_logits, cats, ignore_index=-1)


def masked_policy_terms(flat_logits, legal, actions):
    """log-softmax over legal actions, log-prob of `actions`, mean entropy.  katago_ppo.py:873-888."""
    ml = flat_logits.masked_fill(~legal, float("-inf"))
    lp = torch.log_softmax(ml, dim=-1)
    new_lp = lp.gather(1, actions[:, None]).squeeze(1)
    ent = -(lp.exp() * lp.masked_fill(~legal, 0.0)).sum(dim=-1).mean()
    return new_lp, ent


@dataclass(frozen=True)
class LossWeights:
    lambda_policy: float = 1.0
    lambda_value: float = 1.5
    lambda_score: float = 0.02
    entropy_coeff: float = 0.01
    clip_epsilon: float = 0.2


def ppo_losses(policy_logits, value_logits, score_lead, legal, actions, old_lp, adv,
               value_cats, score_targets, w: LossWeights):
    """The combined minibatch loss of katago_ppo.py:857-924.  Returns a dict of tensors."""
    B = policy_logits.shape[0]
    flat = policy_logits.reshape(B, -1)
    if bool(flat.isnan().any()):
        raise RuntimeError("NaN in raw policy logits from model forward pass")
    if bool((legal.sum(dim=-1) == 0).any()):
        raise RuntimeError("Batch contains samples with zero legal actions in update().")
    new_lp, ent = masked_policy_terms(flat, legal, actions)
    pl = clip_surrogate(new_lp, old_lp, adv, w.clip_epsilon)
    vl = wdl_ce(value_logits, value_cats)
    sl = F.mse_loss(score_lead.squeeze(-1), score_targets)
    total = w.lambda_policy * pl + (w.lambda_value * vl + w.lambda_score * sl) - w.entropy_coeff * ent
    return {"policy_loss": pl, "value_loss": vl, "score_loss": sl, "entropy": ent,
            "total": total, "new_log_probs": new_lp}


# --------------------------------------------------------------------------- GAE

def gae_grid(rewards, values, terminated, next_value, gamma: float, lam: float,
             override=None, lengths=None) -> np.ndarray:
    """(T,N) GAE, numpy, same operation order as gae.py:192-218 / 261-296.

    delta_t = r_t + gamma*nv_t*(1-term_t) - V_t ; A_t = delta_t + (gamma*lam)*(1-term_t)*A_{t+1}
    nv_t = override[t] if finite else (V_{t+1} | next_value at the last step);
    with `lengths` the env's last valid step (lengths[i]-1) also bootstraps from next_value[i]
    (gae.py:119-141).  Arithmetic dtype follows `values` (gae.py:49, 185).
    """
    values = np.asarray(values)
    dt = values.dtype
    rewards = np.asarray(rewards).astype(dt)
    nd = (1.0 - np.asarray(terminated).astype(np.float32)).astype(np.float32)
    next_value = np.asarray(next_value).astype(dt)
    T, N = rewards.shape
    nv = np.zeros_like(values)
    nv[:-1] = values[1:]
    nv[-1] = next_value
    if lengths is not None:
        last = np.clip(np.asarray(lengths).astype(np.int64) - 1, 0, None)
        nv[last, np.arange(N)] = next_value
    if override is not None:
        ov = np.asarray(override).astype(dt)
        nv = np.where(np.isnan(ov), nv, ov)
    g = dt.type(gamma)
    delta = (rewards + (g * nv) * nd) - values
    # `gamma * lam * not_done` is python-float x fp32 tensor -> an fp32 product, promoted afterwards
    decay = (np.float32(gamma * lam) * nd).astype(dt)
    adv = np.empty_like(values)
    last_gae = np.zeros(N, dtype=dt)
    for t in range(T - 1, -1, -1):
        last_gae = (delta[t] + decay[t] * last_gae).astype(dt)
        adv[t] = last_gae
    return adv


def gae_single(rewards, values, terminated, next_value, gamma, lam, override=None) -> np.ndarray:
    """1-D trajectory form of compute_gae (gae.py:8-73)."""
    r = np.asarray(rewards)[:, None]
    v = np.asarray(values)[:, None]
    t = np.asarray(terminated)[:, None]
    ov = None if override is None else np.asarray(override)[:, None]
    nv = np.asarray(next_value).reshape(1)
    return gae_grid(r, v, t, nv, gamma, lam, ov)[:, 0]


def normalize_advantages(adv: torch.Tensor) -> torch.Tensor:
    """katago_ppo.py:797-798 -- torch.std is the UNBIASED (n-1) estimator."""
    if adv.numel() > 1:
        return (adv - adv.mean()) / (adv.std() + 1e-8)
    return adv


# --------------------------------------------------------------------------- optimiser

def clip_and_adam(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, max_norm: float,
                  beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """clip_grad_norm_(max_norm) followed by one torch.optim.Adam step (defaults, no wd).

    katago_ppo.py:929-932.  In-place on the given lists of fp32 tensors.  `step` is the
    1-based step count AFTER this update.  Returns the pre-clip global L2 norm.
    """
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        g = g * coef
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
    return total


# --------------------------------------------------------------------------- one PPO minibatch step

def ppo_minibatch_step(sd: Dict[str, torch.Tensor], num_blocks: int, batch: Dict[str, torch.Tensor],
                       w: LossWeights, opt_state: Optional[dict], lr: float = 2e-4,
                       grad_clip: float = 1.0, momentum: float = 0.1):
    """forward (train-mode BN) + loss + backward + clip + Adam on one minibatch.

    `sd` is updated in place (weights and BN running statistics).  `opt_state` carries
    {"step", "m", "v"} between calls (created when None).  Returns (metrics, opt_state, grads).
    This is the unit the metric "PPO samples/sec" counts.
    """
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    leaves = [sd[k].detach().clone().requires_grad_(True) for k in names]
    live = dict(sd)
    live.update(dict(zip(names, leaves)))
    pol, val, sco = seresnet_forward(live, batch["obs"], num_blocks, train=True,
                                     momentum=momentum, update_running=True)
    for k in sd:  # running stats / counters were updated in `live`
        if "running_" in k or k.endswith("num_batches_tracked"):
            sd[k] = live[k]
    out = ppo_losses(pol, val, sco, batch["legal"], batch["actions"], batch["old_log_probs"],
                     batch["advantages"], batch["value_cats"], batch["score_targets"], w)
    grads = torch.autograd.grad(out["total"], leaves, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for p, g in zip(leaves, grads)]
    if opt_state is None:
        opt_state = {"step": 0, "m": [torch.zeros_like(p) for p in leaves],
                     "v": [torch.zeros_like(p) for p in leaves]}
    opt_state["step"] += 1
    new_params = [p.detach().clone() for p in leaves]
    gnorm = clip_and_adam(new_params, [g.clone() for g in grads], opt_state["m"], opt_state["v"],
                          opt_state["step"], lr, grad_clip)
    for k, p in zip(names, new_params):
        sd[k] = p
    metrics = {k: float(out[k]) for k in ("policy_loss", "value_loss", "score_loss", "entropy", "total")}
    metrics["gradient_norm"] = float(gnorm)
    return metrics, opt_state, dict(zip(names, grads))


# --------------------------------------------------------------------------- synthetic inputs

def board_like_obs(batch: int, seed: int = 0, channels: int = 50) -> torch.Tensor:
    """Observation with the *structure* of real KataGo-mode planes:
    0-27 one-hot piece planes (<= 40 pieces), 28-41 spatially constant hand counts in [0,1],
    42-43 constant scalars, 44-48 constant binary planes, 49 zero.  Produces dead channels,
    exact amax ties and sigma = 0 from the first layer on.
    """
    g = torch.Generator().manual_seed(seed)
    obs = torch.zeros(batch, channels, 9, 9)
    for b in range(batch):
        n_pieces = int(torch.randint(20, 41, (1,), generator=g))
        squares = torch.randperm(81, generator=g)[:n_pieces]
        planes = torch.randint(0, 28, (n_pieces,), generator=g)
        obs[b, planes, squares // 9, squares % 9] = 1.0
        obs[b, 28:42] = (torch.randint(0, 5, (14,), generator=g).float() / 4.0)[:, None, None]
        obs[b, 42:44] = torch.rand(2, generator=g)[:, None, None]
        obs[b, 44:49] = torch.randint(0, 2, (5,), generator=g).float()[:, None, None]
    return obs


def synth_minibatch(batch: int, seed: int = 1234, obs_kind: str = "randn", legal_kind: str = "third",
                    all_ignored: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic PPO minibatch in the style of the reference's scripts/profile_hotpath.py:436-454."""
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(batch, 50, 9, 9, generator=g) if obs_kind == "randn" else board_like_obs(batch, seed)
    legal = torch.zeros(batch, N_ACTIONS, dtype=torch.bool)
    if legal_kind == "third":
        legal[:, : N_ACTIONS // 3] = True
    elif legal_kind == "all":
        legal[:] = True
    else:  # ragged: random ~5% legal, at least one
        legal = torch.rand(batch, N_ACTIONS, generator=g) < 0.05
        legal[torch.arange(batch), torch.randint(0, N_ACTIONS, (batch,), generator=g)] = True
    # a legal action per row
    scores = torch.rand(batch, N_ACTIONS, generator=g).masked_fill(~legal, -1.0)
    actions = scores.argmax(dim=1)
    cats = torch.randint(-1, 3, (batch,), generator=g)
    if all_ignored:
        cats = torch.full((batch,), -1, dtype=torch.int64)
    return {
        "obs": obs, "legal": legal, "actions": actions,
        "old_log_probs": -8.0 + 0.3 * torch.randn(batch, generator=g),
        "advantages": torch.randn(batch, generator=g),
        "value_cats": cats,
        "score_targets": torch.randn(batch, generator=g).clamp(-1.5, 1.5),
    }
