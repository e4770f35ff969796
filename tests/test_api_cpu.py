"""CPU tests of the host-side mirror of keisei.training (no GPU, no compute calls into the HIP library):
registry / params validation and error messages, rollout-buffer contract, the generic update() path against the
reference's own update() result, state_dict contract, scalar-contract models, and that libkeisei_amd.so loads and
exports every symbol include/keisei_amd.h declares."""
import ctypes
import math
import re
import sys
from pathlib import Path

import pytest
import torch

from keisei_amd import _lib
from keisei_amd.training import gae as gae_mod
from keisei_amd.training.algorithm_registry import PPOParams, VALID_ALGORITHMS, validate_algorithm_params
from keisei_amd.training.katago_ppo import (KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer,
                                            _amp_dtype_and_device, compute_value_metrics, ppo_clip_loss,
                                            wdl_cross_entropy_loss)
from keisei_amd.training.model_registry import (VALID_ARCHITECTURES, build_model, get_model_contract,
                                                get_obs_channels, validate_model_params)
from keisei_amd.training.models.katago_base import KataGoBaseModel, KataGoOutput
from keisei_amd.training.models.se_resnet import GlobalPoolBiasBlock, _global_pool
from keisei_amd.training.value_adapter import MultiHeadValueAdapter, ScalarValueAdapter, get_value_adapter
from oracle import keisei_oracle as orc

ROOT = Path(__file__).resolve().parent.parent
TINY = dict(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
            value_fc_size=32, score_fc_size=16, obs_channels=50)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "keisei_amd.h").read_text()
    declared = set(re.findall(r"\b(ka_[a-z0-9_]+)\s*\(", header))
    assert declared, "header declares no entry points?"
    assert _lib.library_path().exists(), "build the library first: python -m keisei_amd.build"
    lib = ctypes.CDLL(str(_lib.library_path()))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"
    assert declared == set(_lib.exported_symbols()), set(_lib.exported_symbols()) ^ declared
    assert _lib.query("ka_version") >= 1


def test_gpu_path_never_falls_back(monkeypatch):
    """With the shared library unavailable the GPU entry points raise; nothing silently runs elsewhere."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_LIB_PATH", ROOT / "keisei_amd" / "does_not_exist.so")
    monkeypatch.setattr(_lib, "_load_error", None)
    with pytest.raises(_lib.KeiseiHipError, match="no fallback"):
        _lib.call("ka_gae", *([None] * 13))
    assert not _lib.available()


# ------------------------------------------------------------------ registries
def test_registry_contracts():
    assert VALID_ARCHITECTURES == {"resnet", "mlp", "transformer", "se_resnet"}
    assert all(get_obs_channels(a) == 50 for a in VALID_ARCHITECTURES)
    assert get_model_contract("se_resnet") == "multi_head" and get_model_contract("mlp") == "scalar"
    assert VALID_ALGORITHMS == {"katago_ppo"} and "ppo" not in VALID_ALGORITHMS
    assert PPOParams().learning_rate == 3e-4
    assert isinstance(validate_algorithm_params("katago_ppo", {"batch_size": 64}), KataGoPPOParams)
    with pytest.raises(ValueError, match="Unknown algorithm 'ppo'"):
        validate_algorithm_params("ppo", {})
    with pytest.raises(TypeError, match="Invalid params for 'katago_ppo'"):
        validate_algorithm_params("katago_ppo", {"nope": 1})
    with pytest.raises(ValueError, match="Unknown architecture 'cnn'"):
        build_model("cnn", {})
    with pytest.raises(TypeError, match="Invalid params for 'se_resnet'"):
        validate_model_params("se_resnet", {"depth": 3})
    with pytest.raises(ValueError, match="num_blocks must be >= 1"):
        validate_model_params("se_resnet", {"num_blocks": 0})
    with pytest.raises(ValueError, match=r"channels \(8\) // se_reduction \(16\)"):
        validate_model_params("se_resnet", {"channels": 8, "se_reduction": 16})
    with pytest.raises(ValueError, match="must be divisible"):
        validate_model_params("transformer", {"d_model": 30, "nhead": 4, "num_layers": 1})
    with pytest.raises(ValueError, match="hidden_sizes"):
        validate_model_params("mlp", {"hidden_sizes": [16, 0]})
    with pytest.raises(ValueError, match="hidden_size must be > 0"):
        validate_model_params("resnet", {"hidden_size": 0, "num_layers": 1})


@pytest.mark.parametrize("field,val,msg", [("batch_size", 0, "batch_size must be > 0"), ("epochs_per_batch", 0, "epochs_per_batch"),
                                           ("gamma", 1.5, "gamma must be in"), ("gae_lambda", -0.1, "gae_lambda"),
                                           ("clip_epsilon", -1.0, "clip_epsilon"), ("learning_rate", 0.0, "learning_rate"),
                                           ("grad_clip", 0.0, "grad_clip")])
def test_ppo_params_validation(field, val, msg):
    with pytest.raises(ValueError, match=msg):
        KataGoPPOParams(**{field: val})


def test_ppo_param_defaults_match_reference():
    p = KataGoPPOParams()
    assert (p.learning_rate, p.gamma, p.gae_lambda, p.clip_epsilon, p.epochs_per_batch, p.batch_size) == (2e-4, 0.99, 0.95, 0.2, 4, 256)
    assert (p.lambda_policy, p.lambda_value, p.lambda_score, p.lambda_entropy, p.score_normalization, p.grad_clip) == \
        (1.0, 1.5, 0.02, 0.01, 76.0, 1.0)
    assert (p.use_amp, p.compile_mode, p.compile_dynamic, p.entropy_decay_epochs, p.score_blend_alpha,
            p.use_terminated_for_gae) == (False, None, True, 0, 0.0, True)


# ------------------------------------------------------------------ model
def test_state_dict_contract_and_cpu_forward_parity(golden):
    g = golden("g2_model_tiny")
    m = build_model("se_resnet", TINY)
    sd = g.sub("sd.")
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    assert isinstance(m, KataGoBaseModel) and isinstance(m.blocks[0], GlobalPoolBiasBlock)
    m.eval()
    with torch.no_grad():
        o = m(g["randn.obs"])
    assert isinstance(o, KataGoOutput) and o.policy_logits.shape == (4, 9, 9, 139)
    assert torch.allclose(o.policy_logits, g["randn.eval.policy"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(o.value_logits, g["randn.eval.value"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(o.score_lead, g["randn.eval.score"], rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        sub = m(g["randn.obs"], gather_idx=torch.tensor([2, 0]))
    assert torch.allclose(sub.value_logits, o.value_logits[[2, 0]], atol=1e-6)
    assert len(list(m.parameters())) == 16 + 14 * 2
    assert any(isinstance(x, torch.nn.BatchNorm2d) for x in m.modules())
    with pytest.raises(ValueError, match=r"Expected obs shape \(batch, 50, 9, 9\), got \(2, 9, 9, 50\)"):
        m(torch.zeros(2, 9, 9, 50))
    pool = _global_pool(torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]]))
    assert torch.allclose(pool, torch.tensor([[2.5, 4.0, math.sqrt(1.25)]]))


def test_configure_amp_freeze():
    m = build_model("se_resnet", TINY)
    m.configure_amp(True, torch.bfloat16, "cpu")
    assert m._amp_enabled and m._amp_dtype == torch.bfloat16
    o = m(torch.randn(2, 50, 9, 9))
    assert o.value_logits.dtype == torch.bfloat16           # CPU autocast path, like the reference
    m._amp_frozen = True
    with pytest.raises(RuntimeError, match="configure_amp"):
        m.configure_amp(False)
    assert _amp_dtype_and_device(True, torch.device("cpu")) == (torch.bfloat16, "cpu")
    assert _amp_dtype_and_device(False, torch.device("cpu"))[1] == "cpu"


def test_scalar_contract_models_match_reference(golden):
    g = golden("g7_scalar")
    obs = g["obs"]
    for arch, p in (("mlp", {"hidden_sizes": [32, 16]}), ("transformer", {"d_model": 32, "nhead": 4, "num_layers": 2}),
                    ("resnet", {"hidden_size": 16, "num_layers": 2})):
        m = build_model(arch, p).eval()
        m.load_state_dict(orc.closed_form_fill(m.state_dict()), strict=True)
        with torch.no_grad():
            pol, val = m(obs)
        assert pol.shape == (3, 11259) and val.shape == (3, 1)
        assert torch.allclose(pol, g[f"{arch}.policy"], rtol=1e-4, atol=1e-5), arch
        assert torch.allclose(val, g[f"{arch}.value"], rtol=1e-4, atol=1e-5), arch
        with pytest.raises(ValueError, match="appears to be NHWC"):
            m(torch.zeros(2, 9, 9, 50))


# ------------------------------------------------------------------ adapters / losses
def test_value_adapters(golden):
    g = golden("g3_loss")
    a = get_value_adapter("multi_head", 1.5, 0.1, 0.1)
    assert isinstance(a, MultiHeadValueAdapter) and isinstance(get_value_adapter("scalar"), ScalarValueAdapter)
    with pytest.raises(ValueError, match="Unknown model contract"):
        get_value_adapter("other")
    with pytest.raises(ValueError, match="score_blend_alpha"):
        MultiHeadValueAdapter(score_blend_alpha=1.5)
    for tag in ("third.", "ragged."):
        vl, sc = g[tag + "value_logits"], g[tag + "score"]
        assert torch.allclose(a.scalar_value_from_output(vl), g[tag + "scalar_value"], atol=1e-6)
        assert torch.allclose(a.scalar_value_blended(vl, sc * 3), g[tag + "scalar_blended"], atol=1e-6)
        loss = a.compute_value_loss(vl, None, g[tag + "value_cats"], g[tag + "score_targets"], sc)
        assert torch.allclose(loss, g[tag + "adapter_loss"], atol=1e-6)
        assert torch.allclose(wdl_cross_entropy_loss(vl, g[tag + "value_cats"]), g[tag + "value_loss"], atol=1e-6)
        pl = ppo_clip_loss(g[tag + "new_log_probs"], g[tag + "old_log_probs"], g[tag + "advantages"], 0.2)
        assert torch.allclose(pl, g[tag + "policy_loss"], atol=1e-6)
    with pytest.raises(ValueError, match="requires value_cats"):
        a.compute_value_loss(torch.zeros(2, 3))
    with pytest.raises(ValueError, match="requires returns"):
        ScalarValueAdapter().compute_value_loss(torch.zeros(2, 1), None)
    vm = compute_value_metrics(torch.tensor([[2.0, 0, 0], [0, 2.0, 0], [0, 0, 2.0], [2.0, 0, 0]]), torch.tensor([0, 1, 2, 1]))
    assert vm == {"value_accuracy": 0.75, "frac_predicted_win": 0.5, "frac_predicted_draw": 0.25, "frac_predicted_loss": 0.25}


# ------------------------------------------------------------------ GAE (host path; bit-exact with the reference)
def test_gae_host_path(golden):
    g = golden("g4_gae")
    r, v, t, nv = g["rewards"], g["values"], g["terminated"], g["next_value"]
    assert torch.equal(gae_mod.compute_gae_gpu(r, v, t.float(), nv, 0.99, 0.95), g["adv_gpu"])
    assert torch.allclose(gae_mod.compute_gae(r, v, t, nv, 0.99, 0.95, next_value_override=g["override"]), g["adv_override"],
                          rtol=1e-5, atol=1e-5)
    assert torch.equal(gae_mod.compute_gae_padded_gpu(r, v, g["terminated_padded"], nv, g["lengths"], 0.99, 0.95), g["adv_padded_gpu"])
    assert torch.allclose(gae_mod.compute_gae_padded(r, v, g["terminated_padded"], nv, g["lengths"], 0.99, 0.95,
                                                     next_value_override=g["override"]), g["adv_padded_override"], rtol=1e-5, atol=1e-5)
    a1 = gae_mod.compute_gae(g["r1"], g["v1"], g["d1"], torch.tensor(0.3), 0.99, 0.95)
    assert a1.shape == (5,) and a1.dtype == torch.float32 and torch.allclose(a1, g["adv1_a"], atol=1e-6)
    a64 = gae_mod.compute_gae(r[:16, :4], v[:16, :4].double(), t[:16, :4], nv[:4].double(), 0.99, 0.95)
    assert a64.dtype == torch.float64 and torch.allclose(a64, g["adv_f64"], rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError, match="only supports 2D"):
        gae_mod.compute_gae_gpu(r[:, 0], v[:, 0], t[:, 0].float(), nv[0], 0.99, 0.95)
    vg = v.clone().requires_grad_(True)
    assert not gae_mod.compute_gae(r, vg, t, nv, 0.99, 0.95).requires_grad


# ------------------------------------------------------------------ buffer
def _step(n=3, **over):
    legal = torch.zeros(n, 11259, dtype=torch.bool)
    legal[:, :5] = True
    d = dict(obs=torch.randn(n, 50, 9, 9), actions=torch.zeros(n, dtype=torch.long), log_probs=torch.zeros(n),
             values=torch.zeros(n), rewards=torch.zeros(n), dones=torch.zeros(n, dtype=torch.bool),
             terminated=torch.zeros(n, dtype=torch.bool), legal_masks=legal,
             value_categories=torch.full((n,), -1), score_targets=torch.zeros(n))
    d.update(over)
    return d


def test_rollout_buffer_contract():
    buf = KataGoRolloutBuffer(num_envs=3, obs_shape=(50, 9, 9), action_space=11259)
    with pytest.raises(ValueError, match="Cannot flatten an empty buffer"):
        buf.flatten()
    buf.add(**_step())
    buf.add(**_step(values=torch.ones(3)))
    assert buf.size == 2 and buf._write_offset == 6
    flat = buf.flatten()
    assert flat["observations"].shape == (6, 50, 9, 9) and flat["legal_masks"].dtype == torch.bool
    assert not flat["observations"].is_cuda and "env_ids" not in flat and "next_value_override" not in flat
    with pytest.raises(AssertionError, match="terminated must be a subset of dones"):
        buf.add(**_step(terminated=torch.ones(3, dtype=torch.bool)))
    with pytest.raises(ValueError, match="invalid values"):
        buf.add(**_step(value_categories=torch.tensor([0, 3, -1])))
    with pytest.raises(ValueError, match="contains NaN"):
        buf.add(**_step(score_targets=torch.tensor([0.0, float("nan"), 0.0])))
    with pytest.raises(ValueError, match="appear unnormalized"):
        buf.add(**_step(score_targets=torch.tensor([0.0, 40.0, 0.0])))
    buf.fill_alternating_perspective_overrides()
    ov = buf.flatten()["next_value_override"].view(2, 3)
    assert torch.equal(ov[0], -torch.ones(3)) and torch.isnan(ov[1]).all()
    buf.clear()
    assert buf.size == 0
    b2 = KataGoRolloutBuffer(2, (50, 9, 9), 11259)
    b2.add(**_step(2), env_ids=torch.tensor([1, 0]), next_value_override=torch.tensor([0.5, float("nan")]))
    f2 = b2.flatten()
    assert f2["env_ids"].tolist() == [1, 0] and f2["next_value_override"][0] == 0.5
    b2.fill_alternating_perspective_overrides()       # no-op for the env_ids layout
    for _ in range(600):                                # growth beyond the initial 512*num_envs rows
        b2.add(**_step(2), env_ids=torch.tensor([1, 0]))
    assert b2._alloc_samples >= 1202 and b2.flatten()["env_ids"].shape == (1202,)


# ------------------------------------------------------------------ algorithm (generic path) vs the reference's update()
def _make_algo(golden, **pp):
    g = golden("g5_update")
    m = build_model("se_resnet", dict(TINY, num_blocks=1))
    m.load_state_dict(g.sub("sd0."))
    params = KataGoPPOParams(learning_rate=1e-3, epochs_per_batch=2, batch_size=8, lambda_score=0.1, score_blend_alpha=0.1, **pp)
    algo = KataGoPPOAlgorithm(params, m)
    buf = KataGoRolloutBuffer(4, (50, 9, 9), 11259)
    d = g.sub("buf.")
    for t in range(4):
        sl = slice(4 * t, 4 * t + 4)
        buf.add(d["observations"][sl], d["actions"][sl], d["log_probs"][sl], d["values"][sl], d["rewards"][sl], d["dones"][sl],
                d["terminated"][sl], d["legal_masks"][sl], d["value_categories"][sl], d["score_targets"][sl],
                next_value_override=d["next_value_override"][sl])
    return g, m, algo, buf


def test_update_matches_reference_on_cpu(golden, monkeypatch):
    g, m, algo, buf = _make_algo(golden)
    it = iter(list(g["perms"]))
    monkeypatch.setattr(torch, "randperm", lambda n, *a, **k: next(it))
    seen = {}
    real = gae_mod.compute_gae

    def spy(*a, **k):
        seen["kw"] = k
        return real(*a, **k)

    monkeypatch.setattr(gae_mod, "compute_gae", spy)       # update() resolves the function from the module at call time
    beats = []
    met = algo.update(buf, g["next_values"], value_adapter=MultiHeadValueAdapter(1.5, 0.1, 0.1), heartbeat_fn=lambda: beats.append(1))
    assert seen["kw"]["next_value_override"].shape == (4, 4)
    assert len(beats) == 4 and buf.size == 0 and m.training
    for k in ("policy_loss", "value_loss", "score_loss", "entropy", "gradient_norm", "value_accuracy", "frac_predicted_loss"):
        assert abs(met[k] - float(g.np("metric." + k))) <= 1e-5 * max(1.0, abs(float(g.np("metric." + k)))), k
    for k, v in g.sub("sd1.").items():
        assert torch.allclose(m.state_dict()[k].float(), v.float(), rtol=1e-4, atol=5e-6), k
    assert algo.timings == {"select_actions_forward_ms": [], "update_forward_backward_ms": [], "gae_ms": []}


def test_algorithm_surface(golden):
    g, m, algo, buf = _make_algo(golden, compile_mode="default", entropy_decay_epochs=10)
    assert algo.compiled_train._orig_mod is algo.forward_model and algo.compiled_eval._orig_mod is algo.forward_model
    assert m._amp_frozen and not algo.scaler.is_enabled() and isinstance(algo.optimizer, torch.optim.Adam)
    algo.warmup_epochs, algo.warmup_entropy = 2, 0.05
    assert algo.get_entropy_coeff(0) == 0.05 and algo.get_entropy_coeff(12) == 0.01
    assert abs(algo.get_entropy_coeff(7) - (0.05 + 0.5 * (0.01 - 0.05))) < 1e-12
    obs = g.sub("buf.")["observations"][:5]
    legal = g.sub("buf.")["legal_masks"][:5]
    actions, lp, vals = algo.select_actions(obs, legal)
    assert actions.shape == (5,) and bool(legal[torch.arange(5), actions].all()) and m.training and vals.abs().max() <= 1
    with pytest.raises(RuntimeError, match=r"Environments \[1\] have zero legal actions"):
        bad = legal.clone()
        bad[1] = False
        algo.select_actions(obs, bad)
    assert torch.allclose(KataGoPPOAlgorithm.scalar_value(torch.tensor([[0.0, 0.0, 0.0]])), torch.zeros(1))
    lin = torch.nn.Linear(1, 1)
    KataGoPPOAlgorithm(KataGoPPOParams(), lin)          # any nn.Module is accepted (no configure_amp)
    with pytest.raises(AssertionError, match="must share parameters"):
        KataGoPPOAlgorithm(KataGoPPOParams(), lin, forward_model=torch.nn.Linear(1, 1))
    buf._storage["legal_masks"][2] = False
    with pytest.raises(RuntimeError, match="zero legal actions in update"):
        algo.update(buf, g["next_values"])


def test_per_env_gae_layout(golden, monkeypatch):
    """env_ids (split-merge) layout routes through compute_gae_padded with lengths as 5th positional argument."""
    g, m, algo, _ = _make_algo(golden)
    buf = KataGoRolloutBuffer(4, (50, 9, 9), 11259)
    d = g.sub("buf.")
    ids = [torch.tensor([0, 1, 2, 3]), torch.tensor([0, 2]), torch.tensor([0, 1, 2, 3]), torch.tensor([3])]
    pos = 0
    for e in ids:
        sl = slice(pos, pos + len(e))
        pos += len(e)
        buf.add(d["observations"][sl], d["actions"][sl], d["log_probs"][sl], d["values"][sl], d["rewards"][sl], d["dones"][sl],
                d["terminated"][sl], d["legal_masks"][sl], d["value_categories"][sl], d["score_targets"][sl], env_ids=e)
    calls = {}
    real = gae_mod.compute_gae_padded

    def spy(*a, **k):
        calls["lengths"] = a[4]
        calls["kw"] = k
        return real(*a, **k)

    monkeypatch.setattr(gae_mod, "compute_gae_padded", spy)
    met = algo.update(buf, g["next_values"])
    assert calls["lengths"].tolist() == [3, 2, 3, 3] and "next_value_override" in calls["kw"]
    assert all(math.isfinite(v) for v in met.values())
    buf.add(d["observations"][:1], d["actions"][:1], d["log_probs"][:1], d["values"][:1], d["rewards"][:1], d["dones"][:1],
            d["terminated"][:1], d["legal_masks"][:1], d["value_categories"][:1], d["score_targets"][:1], env_ids=torch.tensor([9]))
    with pytest.raises(IndexError, match="env_id 9 >= next_values size 4"):
        algo.update(buf, g["next_values"])


def test_rollout_helpers_of_the_loop_on_the_cpu():
    """keisei_amd.training.katago_loop (katago_loop.py:63-431): the reference's call forms on CPU tensors."""
    import numpy as np

    from keisei_amd.training.katago_loop import (PendingTransitions, _compute_value_cats, _negate_where, _resolve_opponent_devices,
                                                  sign_correct_bootstrap, split_merge_step, to_learner_perspective)
    from keisei_amd.training.model_registry import build_model

    torch.manual_seed(0)
    cfg = dict(num_blocks=1, channels=16, se_reduction=4, global_pool_channels=8, policy_channels=8, value_fc_size=16,
               score_fc_size=8, obs_channels=50)
    learner, opp = build_model("se_resnet", dict(cfg)).eval(), build_model("se_resnet", dict(cfg)).eval()
    n = 6
    obs = torch.randn(n, 50, 9, 9)
    legal = torch.zeros(n, 11259, dtype=torch.bool); legal[:, 100:140] = True
    players = np.array([0, 1, 0, 1, 1, 0], dtype=np.uint8)
    sm = split_merge_step(obs, legal, players, learner, opponent_model=opp, learner_side=0)
    assert sm.learner_indices.tolist() == [0, 2, 5] and sm.learner_mask.tolist() == [True, False, True, False, False, True]
    assert bool(legal[torch.arange(n), sm.actions].all()) and sm.learner_log_probs.shape == (3,) and sm.learner_values.shape == (3,)
    sm2 = split_merge_step(obs, legal, players, learner, opponent_models={7: opp, 9: opp}, env_opponent_ids=np.array([7, 9, 7, 9, 7, 9]),
                           learner_side=np.array([0, 0, 1, 1, 0, 0]))
    assert sm2.learner_mask.tolist() == [True, False, False, True, False, True]
    assert _resolve_opponent_devices({0: opp}, torch.device("cpu")) == {0: None}
    v = torch.tensor([1.0, -2.0, 3.0])
    assert _negate_where(v, np.array([True, False, True])).tolist() == [-1.0, -2.0, -3.0] and v.tolist() == [1.0, -2.0, 3.0]
    assert to_learner_perspective(v, np.array([0, 1, 1]), 0).tolist() == [1.0, 2.0, -3.0]
    assert sign_correct_bootstrap(v, np.array([1, 1, 0]), np.array([1, 0, 0])).tolist() == [1.0, 2.0, 3.0]
    cats = _compute_value_cats(torch.tensor([1.0, 0.0, -1.0, 1.0]), torch.tensor([True, True, True, False]), torch.device("cpu"))
    assert cats.tolist() == [0, 1, 2, -1]
    pend = PendingTransitions(3, (2,), 5, torch.device("cpu"))
    m = torch.tensor([True, False, True])
    z = torch.zeros(3)
    pend.create(m, torch.ones(3, 2), torch.tensor([4, 4, 4]), z, z, torch.ones(3, 5, dtype=torch.bool), z, z)
    with pytest.raises(RuntimeError, match="already-valid pending transition"):
        pend.create(m, torch.ones(3, 2), torch.tensor([4, 4, 4]), z, z, torch.ones(3, 5, dtype=torch.bool), z, z)
    pend.accumulate_reward(torch.tensor([1.0, 5.0, -1.0]))
    out = pend.finalize(torch.tensor([True, True, False]), torch.tensor([True, False, False]), torch.tensor([True, False, False]))
    assert out["env_ids"].tolist() == [0] and out["rewards"].tolist() == [1.0] and pend.valid.tolist() == [False, False, True]
    assert pend.finalize(torch.tensor([False, True, False]), torch.zeros(3, dtype=torch.bool), torch.zeros(3, dtype=torch.bool)) is None


def test_bench_spawns_its_own_ranks(monkeypatch, capsys):
    """bench.py --gpus N without a launcher: N children through torch.distributed.run on 127.0.0.1 with a free port, the
    parent's own arguments forwarded (VERDICT r2 'What's missing' 1; reference run.sh:309-310)."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["[Gloo] Rank 0 is connected to 1 peer ranks.\n", '{"metric": "x"}\n'])

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the launcher's status is the parent's
    printed = capsys.readouterr()
    assert printed.out == '{"metric": "x"}\n' and "[Gloo]" in printed.err          # only the JSON line reaches stdout
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_mode_cache_follows_structural_changes():
    """the cached flat module list of _set_training() must not outlive a structural change (a swapped
    submodule used to keep its old train / eval flag for up to 32 calls)."""
    from keisei_amd.training._structure import structure_version
    model = build_model("se_resnet", dict(num_blocks=2, channels=32, se_reduction=8, global_pool_channels=16,
                                          policy_channels=8, value_fc_size=32, score_fc_size=16))
    algo = KataGoPPOAlgorithm(KataGoPPOParams(), model)
    algo._set_training(False)
    assert not any(m.training for m in model.modules())
    v0 = structure_version()
    model.blocks[1].bn1 = torch.nn.BatchNorm2d(32)                        # e.g. a rebuilt / converted layer: born in train mode
    assert structure_version() > v0
    algo._set_training(False)
    assert not any(m.training for m in model.modules())
    algo._set_training(True)
    assert all(m.training for m in model.modules())

    class Odd(torch.nn.Module):                                          # a train() override: the plain recursive call is used
        calls = 0

        def train(self, mode=True):
            Odd.calls += 1
            return super().train(mode)

    model.extra = Odd()
    algo._set_training(False)
    assert Odd.calls >= 1 and not model.extra.training
    v1 = structure_version()
    del model.extra                                                      # removals count too: the cache is rebuilt
    assert structure_version() > v1
    algo._set_training(True)
    assert algo._mode_cache[1] is not None and all(m.training for m in model.modules())
    v2 = structure_version()
    model.blocks[0].bn1.running_mean = None                              # a buffer dropped by assignment
    assert structure_version() > v2
    v3 = structure_version()
    model.blocks[0].bn1.register_buffer("running_mean", torch.zeros(32))
    assert structure_version() > v3
