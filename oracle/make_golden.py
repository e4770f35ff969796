#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference implementation on CPU.

Dev-container only: imports the reference from /root/reference (read-only; bytecode
writes disabled).  The reference never travels to the GPU box -- only the small .npz
fixtures (inputs + expected outputs, fp32) committed under tests/golden/ do.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Fixture inventory:
  g1_block      GlobalPoolBiasBlock(32,8,16): eval/train outputs + input/param grads
  g2_model_tiny SEResNetParams(2,32,8,16,8,32,16,50): state_dict, outputs, grads (randn + board-like obs)
  g2_model_mid16 6x128 and 3x256, init-like closed-form weights (oracle.init_like_state_dict), 16 boards (8 randn + 8
                board-like) chosen among 200 seeds for the widest ReLU margin; fp32 outputs, fp64 gradient norms /
                small tensors / conv slices, and the fp32-vs-fp64 distance of the reference itself
  g2_model_full the headline 40x256 model, same recipe (16 boards): fp32 outputs, gradient norms of all 576 tensors,
                small tensors and conv slices from the fp32 AND the fp64 run of the reference
  g3_loss       masked log-softmax / clip / entropy / CE / MSE terms + gradients wrt logits
  g4_gae        reference-test known answers, (128,64) random w/ dones, NaN override, padded
  g5_update     one full KataGoPPOAlgorithm.update() on CPU with recorded randperm sequences
  g6_adam       clip_grad_norm_ + Adam, 3 steps
  g7_scalar     mlp / transformer tiny forward
  g9_transformer TransformerModel (2 small configs, hash weights): eval / train(dropout 0) outputs, fp64 gradients
  g9_transformer_d256  the same at the benched head shape (d 256, 8 heads, 1 layer, 2 boards): `make_golden.py g9_d256`
  g8_sl         two SLTrainer.train_epoch() calls on a 3-shard directory: shard arrays, visiting order, metrics, weights
"""

from __future__ import annotations

import os
import sys
from pathlib import Path

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("KEISEI_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REF))
sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from keisei.training import gae as ref_gae  # noqa: E402
from keisei.training.katago_ppo import (  # noqa: E402
    KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer, ppo_clip_loss, wdl_cross_entropy_loss,
)
from keisei.training.model_registry import build_model  # noqa: E402
from keisei.training.models.se_resnet import GlobalPoolBiasBlock, SEResNetModel, SEResNetParams, _global_pool  # noqa: E402
from keisei.training.value_adapter import MultiHeadValueAdapter  # noqa: E402

from oracle import keisei_oracle as orc  # noqa: E402

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def npz(name: str, **arrays) -> None:
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(OUT / f"{name}.npz", **conv)
    size = (OUT / f"{name}.npz").stat().st_size
    print(f"  wrote {name}.npz  ({size / 1024:.1f} KiB, {len(conv)} arrays)")


def sd_arrays(prefix: str, sd) -> dict:
    return {f"{prefix}{k}": v.detach().clone() for k, v in sd.items()}


def g1_block() -> None:
    torch.manual_seed(11)
    blk = GlobalPoolBiasBlock(32, 8, 16)
    with torch.no_grad():  # non-trivial BN affine / running stats
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    x = torch.relu(torch.randn(4, 32, 9, 9))  # block inputs are post-ReLU in the model
    x[1, 3] = 0.0          # dead channel: amax tie over all 81, sigma = 0
    x[2, 5] = 0.75         # constant plane: sigma = 0 with non-zero mean
    cot = torch.randn(4, 32, 9, 9)
    arrays = sd_arrays("sd.", blk.state_dict())
    arrays.update(x=x, cot=cot, pool=_global_pool(x))
    blk.eval()
    arrays["out_eval"] = blk(x)
    blk.train()
    for bn in (blk.bn1, blk.bn2):
        bn.momentum = 0.0
    xr = x.clone().requires_grad_(True)
    out = blk(xr)
    arrays["out_train"] = out
    grads = torch.autograd.grad((out * cot).sum(), [xr] + list(blk.parameters()))
    arrays["grad.x"] = grads[0]
    for (n, _), g in zip(blk.named_parameters(), grads[1:]):
        arrays["grad." + n] = g
    npz("g1_block", **arrays)


def _model_case(model, obs, prefix, arrays, cot_seed):
    g = torch.Generator().manual_seed(cot_seed)
    model.eval()
    with torch.no_grad():
        o = model(obs)
    arrays[prefix + "eval.policy"] = o.policy_logits.contiguous()
    arrays[prefix + "eval.value"] = o.value_logits
    arrays[prefix + "eval.score"] = o.score_lead
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0
    o = model(obs)
    arrays[prefix + "train.policy"] = o.policy_logits.contiguous()
    arrays[prefix + "train.value"] = o.value_logits
    arrays[prefix + "train.score"] = o.score_lead
    cp = torch.randn(o.policy_logits.shape, generator=g)
    cv = torch.randn(o.value_logits.shape, generator=g)
    cs = torch.randn(o.score_lead.shape, generator=g)
    arrays[prefix + "cot.policy"], arrays[prefix + "cot.value"], arrays[prefix + "cot.score"] = cp, cv, cs
    loss = (o.policy_logits * cp).sum() / obs.shape[0] + (o.value_logits * cv).sum() + (o.score_lead * cs).sum()
    grads = torch.autograd.grad(loss, list(model.parameters()))
    return dict(zip([n for n, _ in model.named_parameters()], grads))


def g2_model_tiny() -> None:
    torch.manual_seed(22)
    params = SEResNetParams(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16,
                            policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)
    model = SEResNetModel(params)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
    arrays = sd_arrays("sd.", model.state_dict())
    obs_a = torch.randn(4, 50, 9, 9)
    obs_b = orc.board_like_obs(4, seed=5)
    arrays["randn.obs"], arrays["board.obs"] = obs_a, obs_b
    for tag, obs, seed in (("randn.", obs_a, 1), ("board.", obs_b, 2)):
        grads = _model_case(model, obs, tag, arrays, seed)
        for n, gval in grads.items():
            arrays[f"{tag}grad.{n}"] = gval
    npz("g2_model_tiny", **arrays)


def _ref_model(shape, sd, dtype):
    model = SEResNetModel(SEResNetParams(**shape.__dict__))
    model.load_state_dict(sd, strict=True)
    model = model.to(dtype)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0
    return model


def _relu_margin(shape, sd, obs):
    """(smallest |ReLU input| of a train-mode fp64 forward of the reference, number of ReLU inputs)."""
    model = _ref_model(shape, sd, torch.float64).train()
    real_relu, real_frelu = torch.relu, torch.nn.functional.relu
    seen = {"min": float("inf"), "n": 0}

    def rec(x, *a, **kw):
        nz = x.detach().abs()
        nz = nz[nz > 0]                     # exact zeros (dead planes of board-like inputs) are structural, not knife edges
        if nz.numel():
            seen["min"] = min(seen["min"], float(nz.min()))
        seen["n"] += x.numel()
        return real_relu(x)

    torch.relu = lambda x: rec(x)
    torch.nn.functional.relu = rec
    try:
        with torch.no_grad():
            model(obs.double())
    finally:
        torch.relu, torch.nn.functional.relu = real_relu, real_frelu
    return seen["min"], seen["n"]


def _mixed_obs(batch, seed):
    half = batch // 2
    return torch.cat([torch.randn(half, 50, 9, 9, generator=torch.Generator().manual_seed(seed)),
                      orc.board_like_obs(batch - half, seed=seed + 1)])


SMALL = 20000          # gradient tensors up to this many elements are stored whole


def _grad_case(shape, sd, obs, tag, arrays, conv_every):
    """fp32 outputs (eval + train) and gradients of the reference in fp32 and fp64 for closed-form cotangents."""
    B = obs.shape[0]
    cp, cv, cs = orc.closed_form_cotangents(B)
    res = {}
    for dt in (torch.float32, torch.float64):
        model = _ref_model(shape, sd, dt)
        if dt == torch.float32:
            model.eval()
            with torch.no_grad():
                o = model(obs)
            arrays[tag + "eval.policy"], arrays[tag + "eval.value"], arrays[tag + "eval.score"] = \
                o.policy_logits.contiguous(), o.value_logits, o.score_lead
        model.train()
        o = model(obs.to(dt))
        loss = ((o.policy_logits * cp.to(dt)).sum() / B + (o.value_logits * cv.to(dt)).sum()
                + (o.score_lead * cs.to(dt)).sum())
        grads = torch.autograd.grad(loss, list(model.parameters()))
        res[dt] = (o, dict(zip([n for n, _ in model.named_parameters()], grads)))
    o32, g32 = res[torch.float32]
    o64, g64 = res[torch.float64]
    arrays[tag + "train.policy"], arrays[tag + "train.value"], arrays[tag + "train.score"] = \
        o32.policy_logits.contiguous(), o32.value_logits, o32.score_lead
    arrays[tag + "train64.policy_maxdiff"] = np.float64((o32.policy_logits.double() - o64.policy_logits).abs().max())
    names = list(g64.keys())
    arrays[tag + "grad_names"] = np.array(names)
    arrays[tag + "grad_norms64"] = np.array([float(g64[n].norm()) for n in names])
    arrays[tag + "grad_norms32"] = np.array([float(g32[n].double().norm()) for n in names])
    # how far the reference's own fp32 gradients are from its fp64 gradients (ReLU knife edges): the yardstick of the tests
    arrays[tag + "grad_rel32v64"] = np.array([float((g32[n].double() - g64[n]).norm() / (g64[n].norm() + 1e-300))
                                              for n in names])
    convs = [n for n in names if n.endswith(("conv1.weight", "conv2.weight")) and n.startswith("blocks.")]
    keep = set(convs[::conv_every]) | {"input_conv.weight"}
    for n in names:
        if g64[n].numel() <= SMALL:
            arrays[f"{tag}grad64.{n}"] = g64[n].float()
        elif n in keep:
            arrays[f"{tag}grad64.{n}[:2]"] = g64[n][:2].float()
    # the reference's own bf16 mode (CPU autocast, katago_base.py:68-75 / katago_ppo.py:23-24) against its fp64 run:
    # the yardstick for this build's bf16 mode (different op sets are cast, so never bitwise comparable)
    model = _ref_model(shape, sd, torch.float32)
    model.configure_amp(True, torch.bfloat16, "cpu")
    model.eval()
    with torch.no_grad():
        ob = model(obs)
    arrays[tag + "bf16ref.eval.policy_maxdiff"] = np.float64((ob.policy_logits.double() - arrays[tag + "eval.policy"].double()).abs().max())
    model.train()
    ob = model(obs)
    arrays[tag + "bf16ref.train.policy_maxdiff"] = np.float64((ob.policy_logits.detach().double() - o64.policy_logits.detach()).abs().max())
    loss = ((ob.policy_logits.float() * cp).sum() / B + (ob.value_logits.float() * cv).sum() + (ob.score_lead.float() * cs).sum())
    gb = dict(zip(names, torch.autograd.grad(loss, list(model.parameters()))))
    arrays[tag + "bf16ref.grad_rel_v64"] = np.array([float((gb[n].double() - g64[n]).norm() / (g64[n].norm() + 1e-300))
                                                    for n in names])
    arrays[tag + "bf16ref.grad_norms"] = np.array([float(gb[n].double().norm()) for n in names])
    rb = arrays[tag + "bf16ref.grad_rel_v64"]
    print(f"    {tag} reference bf16 autocast (CPU) vs its fp64: policy eval {float(arrays[tag + 'bf16ref.eval.policy_maxdiff']):.3f} "
          f"train {float(arrays[tag + 'bf16ref.train.policy_maxdiff']):.3f} (|logit|max {float(o64.policy_logits.abs().max()):.2f}), "
          f"gradient rel L2 worst {rb.max():.3f} median {float(np.median(rb)):.3f}, "
          f"norm ratio off by {float(np.abs(arrays[tag + 'bf16ref.grad_norms'] / arrays[tag + 'grad_norms64'] - 1).max()):.3f}")
    worst = max(arrays[tag + "grad_rel32v64"])
    print(f"    {tag} fp32-vs-fp64 of the reference: policy {float(arrays[tag + 'train64.policy_maxdiff']):.2e}, "
          f"worst gradient rel L2 {worst:.2e}, median {float(np.median(arrays[tag + 'grad_rel32v64'])):.2e}")


def g2_model_mid16() -> None:
    arrays = {}
    for tag, shape in (("s6x128.", orc.NetShape(num_blocks=6, channels=128)),
                       ("s3x256.", orc.NetShape(num_blocks=3, channels=256))):
        sd = orc.init_like_state_dict(shape)
        best = (-1.0, None)
        for seed in range(1000, 1400, 2):
            margin, n = _relu_margin(shape, sd, _mixed_obs(16, seed))
            if margin > best[0]:
                best = (margin, seed)
        margin, seed = best
        print(f"    {tag} obs seed {seed}: smallest |ReLU input| {margin:.2e} over {n} ReLU inputs (200 seeds tried)")
        obs = _mixed_obs(16, seed)
        arrays[tag + "obs"] = obs
        arrays[tag + "relu_margin"] = np.float64(margin)
        _grad_case(shape, sd, obs, tag, arrays, conv_every=1)
    npz("g2_model_mid16", **arrays)


def g2_model_full() -> None:
    arrays = {}
    shape = orc.NetShape()            # 40 x 256, keisei-katago.toml:15-23
    sd = orc.init_like_state_dict(shape)
    obs = _mixed_obs(16, 4000)
    arrays["obs"] = obs
    margin, n = _relu_margin(shape, sd, obs)
    arrays["relu_margin"] = np.float64(margin)
    print(f"    40x256: smallest |ReLU input| {margin:.2e} over {n} ReLU inputs")
    _grad_case(shape, sd, obs, "", arrays, conv_every=8)
    npz("g2_model_full", **arrays)


def g3_loss() -> None:
    arrays = {}
    for tag, legal_kind, all_ignored in (("third.", "third", False), ("ragged.", "ragged", True)):
        mb = orc.synth_minibatch(4, seed=31 if tag == "third." else 32, legal_kind=legal_kind,
                                 all_ignored=all_ignored)
        g = torch.Generator().manual_seed(33)
        logits = (2.0 * torch.randn(4, 9, 9, 139, generator=g)).requires_grad_(True)
        vlog = torch.randn(4, 3, generator=g).requires_grad_(True)
        score = torch.randn(4, 1, generator=g).requires_grad_(True)
        # -- the reference's own statements, katago_ppo.py:858-924 (inline path, no adapter)
        flat = logits.reshape(4, -1)
        masked = flat.masked_fill(~mb["legal"], float("-inf"))
        lp_all = torch.nn.functional.log_softmax(masked, dim=-1)
        new_lp = lp_all.gather(1, mb["actions"].unsqueeze(1)).squeeze(1)
        old_lp = (new_lp.detach() + 0.3 * torch.randn(4, generator=g))
        pl = ppo_clip_loss(new_lp, old_lp, mb["advantages"], 0.2)
        probs = lp_all.exp()
        ent = -(probs * lp_all.masked_fill(~mb["legal"], 0.0)).sum(dim=-1).mean()
        vl = wdl_cross_entropy_loss(vlog, mb["value_cats"])
        sl = torch.nn.functional.mse_loss(score.squeeze(-1), mb["score_targets"])
        total = 1.0 * pl + (1.5 * vl + 0.1 * sl) - 0.01 * ent
        gl, gv, gs = torch.autograd.grad(total, [logits, vlog, score])
        adapter = MultiHeadValueAdapter(1.5, 0.1, 0.1)
        arrays.update({
            tag + "logits": logits, tag + "value_logits": vlog, tag + "score": score,
            tag + "legal": mb["legal"], tag + "actions": mb["actions"], tag + "old_log_probs": old_lp,
            tag + "advantages": mb["advantages"], tag + "value_cats": mb["value_cats"],
            tag + "score_targets": mb["score_targets"],
            tag + "new_log_probs": new_lp, tag + "policy_loss": pl, tag + "entropy": ent,
            tag + "value_loss": vl, tag + "score_loss": sl, tag + "total": total,
            tag + "adapter_loss": adapter.compute_value_loss(vlog, None, mb["value_cats"], mb["score_targets"], score),
            tag + "scalar_value": adapter.scalar_value_from_output(vlog),
            tag + "scalar_blended": adapter.scalar_value_blended(vlog, score * 3),
            tag + "grad.logits": gl, tag + "grad.value_logits": gv, tag + "grad.score": gs,
        })
    npz("g3_loss", **arrays)


def g4_gae() -> None:
    arrays = {}
    g = torch.Generator().manual_seed(123)
    T, N = 128, 64
    rewards = 0.1 * torch.randn(T, N, generator=g)
    values = torch.randn(T, N, generator=g)
    term = torch.rand(T, N, generator=g) < 0.05
    nv = torch.randn(N, generator=g)
    arrays.update(rewards=rewards, values=values, terminated=term, next_value=nv)
    arrays["adv_loop"] = ref_gae.compute_gae(rewards, values, term, nv, 0.99, 0.95)
    arrays["adv_gpu"] = ref_gae.compute_gae_gpu(rewards, values, term.float(), nv, 0.99, 0.95)
    ov = torch.full((T, N), float("nan"))
    pick = torch.rand(T, N, generator=g) < 0.1
    ov[pick] = torch.randn(int(pick.sum()), generator=g)
    arrays["override"] = ov
    arrays["adv_override"] = ref_gae.compute_gae(rewards, values, term, nv, 0.99, 0.95, next_value_override=ov)
    arrays["adv_override_gpu"] = ref_gae.compute_gae_gpu(rewards, values, term.float(), nv, 0.99, 0.95,
                                                         next_value_override=ov)
    # alternating-perspective pattern (katago_ppo.py:320-362): override = -V[t+1] on non-terminal cells
    alt = torch.full((T, N), float("nan"))
    alt[:-1] = torch.where(term[:-1], torch.tensor(float("nan")), -values[1:])
    arrays["override_alt"] = alt
    arrays["adv_override_alt"] = ref_gae.compute_gae(rewards, values, term, nv, 0.99, 0.95, next_value_override=alt)
    # padded: ragged lengths, padding terminated = 1
    lengths = torch.randint(1, T + 1, (N,), generator=g)
    lengths[0], lengths[1] = T, 1
    term_p = term.float().clone()
    for i in range(N):
        term_p[int(lengths[i]):, i] = 1.0
    arrays["lengths"] = lengths
    arrays["terminated_padded"] = term_p
    arrays["adv_padded"] = ref_gae.compute_gae_padded(rewards, values, term_p, nv, lengths, 0.99, 0.95)
    arrays["adv_padded_gpu"] = ref_gae.compute_gae_padded_gpu(rewards, values, term_p, nv, lengths, 0.99, 0.95)
    arrays["adv_padded_override"] = ref_gae.compute_gae_padded(rewards, values, term_p, nv, lengths, 0.99, 0.95,
                                                               next_value_override=ov)
    # 1-D trajectory, edge parameters
    r1 = torch.tensor([1.0, 2.0, 3.0, -1.0, 0.5])
    v1 = torch.tensor([0.5, 0.4, 0.3, 0.2, 0.1])
    d1 = torch.tensor([False, False, True, False, False])
    arrays.update(r1=r1, v1=v1, d1=d1)
    for tag, (gm, lm) in {"a": (0.99, 0.95), "b": (1.0, 1.0), "c": (0.9, 0.0), "d": (0.0, 0.5)}.items():
        arrays["adv1_" + tag] = ref_gae.compute_gae(r1, v1, d1, torch.tensor(0.3), gm, lm)
    # float64 values -> float64 arithmetic (gae.py:49)
    arrays["adv_f64"] = ref_gae.compute_gae(rewards[:16, :4], values[:16, :4].double(), term[:16, :4],
                                            nv[:4].double(), 0.99, 0.95)
    npz("g4_gae", **arrays)


def _fill_buffer(buf, T, N, gen, with_override):
    for t in range(T):
        legal = torch.rand(N, 11259, generator=gen) < 0.02
        acts = torch.randint(0, 11259, (N,), generator=gen)
        legal[torch.arange(N), acts] = True
        last = t == T - 1
        dones = torch.full((N,), last)
        cats = torch.randint(0, 3, (N,), generator=gen) if last else torch.full((N,), -1)
        buf.add(obs=torch.randn(N, 50, 9, 9, generator=gen), actions=acts,
                log_probs=-5.0 + 0.2 * torch.randn(N, generator=gen), values=0.3 * torch.randn(N, generator=gen),
                rewards=0.1 * torch.randn(N, generator=gen), dones=dones, terminated=dones.clone(),
                legal_masks=legal, value_categories=cats,
                score_targets=torch.randn(N, generator=gen).clamp(-1.5, 1.5),
                next_value_override=(torch.full((N,), float("nan")) if with_override else None))


def g5_update() -> None:
    torch.manual_seed(55)
    mparams = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16,
                   policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)
    model = build_model("se_resnet", mparams)
    arrays = sd_arrays("sd0.", model.state_dict())
    T, N = 4, 4
    pp = KataGoPPOParams(learning_rate=1e-3, epochs_per_batch=2, batch_size=8, lambda_score=0.1,
                         score_blend_alpha=0.1)
    algo = KataGoPPOAlgorithm(pp, model)
    adapter = MultiHeadValueAdapter(pp.lambda_value, pp.lambda_score, pp.score_blend_alpha)
    buf = KataGoRolloutBuffer(num_envs=N, obs_shape=(50, 9, 9), action_space=11259)
    gen = torch.Generator().manual_seed(56)
    _fill_buffer(buf, T, N, gen, with_override=True)
    buf.fill_alternating_perspective_overrides()
    flat = buf.flatten()
    for k, v in flat.items():
        arrays["buf." + k] = v.clone()
    next_values = 0.3 * torch.randn(N, generator=gen)
    arrays["next_values"] = next_values
    perms = []
    real_randperm = torch.randperm

    def recording_randperm(n, *a, **kw):
        p = real_randperm(n, *a, **kw)
        perms.append(p.clone())
        return p

    torch.randperm = recording_randperm
    try:
        metrics = algo.update(buf, next_values, value_adapter=adapter)
    finally:
        torch.randperm = real_randperm
    arrays["perms"] = torch.stack(perms)
    for k, v in metrics.items():
        arrays["metric." + k] = np.float64(v)
    arrays.update(sd_arrays("sd1.", model.state_dict()))
    opt = algo.optimizer.state_dict()["state"]
    arrays["opt.step"] = np.float64(float(opt[0]["step"]))
    arrays["opt.exp_avg.0"] = opt[0]["exp_avg"]
    arrays["opt.exp_avg_sq.0"] = opt[0]["exp_avg_sq"]
    arrays["hyper"] = np.array([pp.learning_rate, pp.gamma, pp.gae_lambda, pp.clip_epsilon, pp.lambda_policy,
                                pp.lambda_value, pp.lambda_score, pp.lambda_entropy, pp.grad_clip,
                                pp.epochs_per_batch, pp.batch_size, T, N])
    npz("g5_update", **arrays)


def g6_adam() -> None:
    g = torch.Generator().manual_seed(66)
    shapes = [(7,), (5, 3), (2, 3, 3, 3), (1,), (64, 9)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    opt = torch.optim.Adam(params, lr=2e-4)
    arrays = {f"p0.{i}": p.detach().clone() for i, p in enumerate(params)}
    for step in range(3):
        scale = [3.0, 0.05, 1.0][step]  # step 0 clips hard, step 1 does not clip
        for i, p in enumerate(params):
            p.grad = scale * torch.randn(p.shape, generator=g)
            arrays[f"g{step}.{i}"] = p.grad.clone()
        norm = torch.nn.utils.clip_grad_norm_(params, 1.0)
        arrays[f"norm{step}"] = norm
        opt.step()
        for i, p in enumerate(params):
            arrays[f"p{step + 1}.{i}"] = p.detach().clone()
    st = opt.state_dict()["state"]
    for i in range(len(params)):
        arrays[f"m.{i}"] = st[i]["exp_avg"]
        arrays[f"v.{i}"] = st[i]["exp_avg_sq"]
    npz("g6_adam", **arrays)


def g7_scalar() -> None:
    """Scalar-contract models: weights from the closed-form generator (policy_fc alone is
    MBs), so only obs + outputs are stored."""
    arrays = {}
    obs = torch.randn(3, 50, 9, 9, generator=torch.Generator().manual_seed(70))
    arrays["obs"] = obs
    for arch, p in (("mlp", {"hidden_sizes": [32, 16]}), ("transformer", {"d_model": 32, "nhead": 4, "num_layers": 2}),
                    ("resnet", {"hidden_size": 16, "num_layers": 2})):
        m = build_model(arch, p).eval()
        m.load_state_dict(orc.closed_form_fill(m.state_dict()), strict=True)
        with torch.no_grad():
            pol, val = m(obs)
        arrays[f"{arch}.policy"], arrays[f"{arch}.value"] = pol, val
    npz("g7_scalar", **arrays)


def g9_transformer(which: str = "small") -> None:
    """TransformerModel (transformer.py:37-95) with init-like hash weights: eval outputs, and -- with every dropout of the
    encoder layers set to p = 0, so that train mode is deterministic -- train-mode outputs and fp64 gradients (norms of
    every tensor, every tensor of <= 20 000 elements in full, 8 rows of the policy matrix).  `which="d256"`: the head shape
    bench.py's transformer workload runs (d 256, 8 heads; one layer, two boards: the 233 M-parameter policy layer is
    hash-filled on both sides, never stored) into its own file g9_transformer_d256.npz."""
    arrays = {}
    cases = ((("d32h4L2.", {"d_model": 32, "nhead": 4, "num_layers": 2}, 3), ("d64h2L1.", {"d_model": 64, "nhead": 2, "num_layers": 1}, 5))
             if which == "small" else (("d256h8L1.", {"d_model": 256, "nhead": 8, "num_layers": 1}, 2),))
    for tag, p, batch in cases:
        m = build_model("transformer", p)
        sd = orc.hash_fill(m.state_dict())
        m.load_state_dict(sd, strict=True)
        obs = _mixed_obs(batch + batch % 2, 900)[:batch]
        arrays[tag + "obs"] = obs
        m.eval()
        with torch.no_grad():
            pol, val = m(obs)
        arrays[tag + "eval.policy"], arrays[tag + "eval.value"] = pol, val
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        cp = orc._hash_uniform(batch * 11259, 7101).float().reshape(batch, 11259)
        cv = orc._hash_uniform(batch, 7102).float().reshape(batch, 1)
        m64 = build_model("transformer", p).double()
        m64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()})
        for mod in m64.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        m.train(); m64.train()
        pol, val = m(obs)
        arrays[tag + "train.policy"], arrays[tag + "train.value"] = pol, val
        pol64, val64 = m64(obs.double())
        loss = (pol64 * cp.double()).sum() / batch + (val64 * cv.double()).sum()
        grads = dict(zip([n for n, _ in m64.named_parameters()], torch.autograd.grad(loss, list(m64.parameters()))))
        names = list(grads)
        arrays[tag + "grad_names"] = np.array(names)
        arrays[tag + "grad_norms64"] = np.array([float(g.norm()) for g in grads.values()])
        for n, g in grads.items():
            if g.numel() <= SMALL:
                arrays[f"{tag}grad64.{n}"] = g.float()
        arrays[tag + "grad64.policy_fc.weight[:8]"] = grads["policy_fc.weight"][:8].float()
    npz("g9_transformer" if which == "small" else "g9_transformer_d256", **arrays)


def g8_sl() -> None:
    """keisei/sl: write_shard -> SLDataset -> SLTrainer.train_epoch() x 2 (CPU fp32), with the order in which the
    shuffling DataLoader visited the positions recorded per epoch."""
    import tempfile

    from keisei.sl import dataset as ref_ds
    from keisei.sl.trainer import SLConfig, SLTrainer

    rng = np.random.default_rng(88)
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        for shard, n in ((0, 9), (1, 6), (2, 7)):
            obs = (rng.random((n, ref_ds.OBS_SIZE)) < 0.1).astype(np.float32) + 0.05 * rng.standard_normal((n, ref_ds.OBS_SIZE)).astype(np.float32)
            pol, val = rng.integers(0, 11259, n), rng.integers(0, 3, n)
            sc = np.clip(rng.standard_normal(n), -2, 2).astype(np.float32)
            ref_ds.write_shard(tmp / f"shard_{shard}.bin", obs, pol, val, sc)
            arrays[f"shard{shard}.obs"], arrays[f"shard{shard}.policy"] = obs, pol
            arrays[f"shard{shard}.value"], arrays[f"shard{shard}.score"] = val, sc
            arrays[f"shard{shard}.bytes"] = np.frombuffer((tmp / f"shard_{shard}.bin").read_bytes(), dtype=np.uint8)
        torch.manual_seed(81)
        mparams = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16,
                       policy_channels=8, value_fc_size=32, score_fc_size=16, obs_channels=50)
        model = build_model("se_resnet", mparams)
        arrays.update(sd_arrays("sd0.", model.state_dict()))
        cfg = SLConfig(data_dir=str(tmp), batch_size=8, learning_rate=1e-3, total_epochs=5, lambda_score=0.05)
        trainer = SLTrainer(model, cfg)
        visited = []
        real_get = ref_ds.SLDataset.__getitem__

        def recording_get(self, idx):
            visited.append(int(idx))
            return real_get(self, idx)

        ref_ds.SLDataset.__getitem__ = recording_get
        try:
            torch.manual_seed(82)
            for ep in range(2):
                visited.clear()
                m = trainer.train_epoch()
                arrays[f"order{ep}"] = np.array(visited, dtype=np.int64)
                for k, v in m.items():
                    arrays[f"metric{ep}.{k}"] = np.float64(v)
                arrays[f"lr{ep}"] = np.float64(trainer.optimizer.param_groups[0]["lr"])
                arrays.update(sd_arrays(f"sd{ep + 1}.", model.state_dict()))
        finally:
            ref_ds.SLDataset.__getitem__ = real_get
        arrays["hyper"] = np.array([cfg.batch_size, cfg.learning_rate, cfg.total_epochs, cfg.lambda_policy, cfg.lambda_value,
                                    cfg.lambda_score, cfg.grad_clip])
    npz("g8_sl", **arrays)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if len(sys.argv) > 1 and sys.argv[1] == "g9_d256":
        g9_transformer("d256")
        print("wrote g9_transformer_d256")
        sys.exit(0)
    for fn in (g1_block, g2_model_tiny, g2_model_mid16, g2_model_full, g3_loss, g4_gae, g5_update, g6_adam, g7_scalar, g8_sl, g9_transformer):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn()
