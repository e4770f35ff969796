"""GPU: the device-resident rollout store against the reference-shaped host store, bit for bit, and
update() from it against the reference's own update() result (golden g5)."""
import numpy as np
import pytest
import torch

from keisei_amd import _lib
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer
from keisei_amd.training.model_registry import build_model
from keisei_amd.training.value_adapter import MultiHeadValueAdapter

pytestmark = pytest.mark.gpu
DEV = "cuda"
A = 11259
MP = dict(num_blocks=1, channels=32, se_reduction=8, global_pool_channels=16, policy_channels=8,
          value_fc_size=32, score_fc_size=16, obs_channels=50)


def st():
    return _lib.stream_ptr()


def same(a, b):
    a, b = a.cpu(), b.cpu()
    if a.dtype.is_floating_point:
        return a.shape == b.shape and bool(((a == b) | (a.isnan() & b.isnan())).all())
    return torch.equal(a, b)


@pytest.mark.parametrize("rows,width", [(7, A), (3, 64), (5, 45), (1, 1), (4, 33)])
def test_mask_bits_round_trip(rows, width):
    g = torch.Generator().manual_seed(rows * 1000 + width)
    legal = torch.rand(rows, width, generator=g) < 0.3
    legal[0] = False
    legal[-1] = True
    words = _lib.query("ka_mask_words", width)
    assert words == (width + 31) // 32
    bits = torch.full((rows, words), -1, dtype=torch.int32, device=DEV)
    _lib.call("ka_pack_mask_bits", legal.to(DEV), bits, rows, width, st())
    padded = np.zeros((rows, words * 32), dtype=np.uint8)
    padded[:, :width] = legal.numpy()
    ref = np.packbits(padded, axis=1, bitorder="little").view("<u4")
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), ref)
    idx = torch.randperm(rows, generator=g)
    out = torch.empty(rows, width, dtype=torch.bool, device=DEV)
    _lib.call("ka_unpack_mask_bits", bits, idx.to(DEV), out, rows, width, st())
    assert torch.equal(out.cpu(), legal[idx])


def synth_steps(T, N, seed, env_layout=False, overrides=True):
    g = torch.Generator().manual_seed(seed)
    steps = []
    for t in range(T):
        n = N if not env_layout else int(torch.randint(1, N + 1, (1,), generator=g))
        done = torch.rand(n, generator=g) < 0.2
        term = done & (torch.rand(n, generator=g) < 0.5)
        legal = torch.rand(n, A, generator=g) < 0.05
        legal[:, 17] = True
        step = dict(obs=torch.randn(n, 50, 9, 9, generator=g), actions=torch.randint(0, A, (n,), generator=g),
                    log_probs=-8 + torch.randn(n, generator=g), values=torch.randn(n, generator=g),
                    rewards=torch.randn(n, generator=g), dones=done.float() if t % 2 else done, terminated=term,
                    legal_masks=legal, value_categories=torch.randint(-1, 3, (n,), generator=g),
                    score_targets=torch.randn(n, generator=g).clamp(-2, 2))
        if env_layout:
            step["env_ids"] = torch.randperm(N, generator=g)[:n]
        if overrides and t % 3 != 1:
            ov = torch.randn(n, generator=g)
            ov[torch.rand(n, generator=g) < 0.6] = float("nan")
            step["next_value_override"] = ov
        steps.append(step)
    return steps


def fill(buf, steps, device):
    for s in steps:
        buf.add(s["obs"].to(device), s["actions"].to(device), s["log_probs"].to(device), s["values"].to(device),
                s["rewards"].to(device), s["dones"].to(device), s["terminated"].to(device), s["legal_masks"].to(device),
                s["value_categories"].to(device), s["score_targets"].to(device),
                env_ids=None if "env_ids" not in s else s["env_ids"].to(device),
                next_value_override=None if "next_value_override" not in s else s["next_value_override"].to(device))


@pytest.mark.parametrize("env_layout", [False, True])
def test_device_store_equals_host_store(env_layout):
    T, N = 7, 5
    steps = synth_steps(T, N, seed=4 + env_layout, env_layout=env_layout)
    host = KataGoRolloutBuffer(N, (50, 9, 9), A)
    dev = KataGoRolloutBuffer(N, (50, 9, 9), A)
    fill(host, steps, "cpu")
    fill(dev, steps, DEV)
    assert dev.is_device_resident and not host.is_device_resident and dev.size == host.size == T
    if not env_layout:
        host.fill_alternating_perspective_overrides()
        dev.fill_alternating_perspective_overrides()
    fh, fd = host.flatten(), dev.flatten()
    assert set(fh) == set(fd)
    for k in fh:
        assert fd[k].is_cuda and fd[k].dtype == fh[k].dtype and same(fd[k], fh[k]), k
    packed = dev.flatten_packed()
    assert "legal_masks" not in packed and packed["legal_bits"].shape == (fh["rewards"].numel(), (A + 31) // 32)
    # a second epoch after clear(): stale override cells must not leak, growth keeps the rows
    host.clear(); dev.clear()
    more = synth_steps(3, N, seed=40, env_layout=env_layout, overrides=False)
    fill(host, more, "cpu"); fill(dev, more, DEV)
    fh, fd = host.flatten(), dev.flatten()
    assert set(fh) == set(fd) and all(same(fd[k], fh[k]) for k in fh)


def test_device_store_takes_packed_masks_as_they_are():
    """add() with the legal masks as PACKED int32 rows (the device env's StepResult.legal_mask_bits, PendingTransitions.finalize()
    ["legal_mask_bits"]): ka_rollout_append_packed copies the words -- the store equals the one filled from bool rows."""
    T, N = 5, 6
    steps = synth_steps(T, N, seed=21)
    ref, dev = KataGoRolloutBuffer(N, (50, 9, 9), A), KataGoRolloutBuffer(N, (50, 9, 9), A)
    fill(ref, steps, DEV)
    words = (A + 31) // 32
    for s in steps:
        legal = s["legal_masks"].to(DEV)
        bits = torch.empty(legal.shape[0], words, dtype=torch.int32, device=DEV)
        _lib.call("ka_pack_mask_bits", legal, bits, legal.shape[0], A, st())
        packed_step = dict(s)
        packed_step["legal_masks"] = bits
        fill(dev, [packed_step], DEV)
    fr, fd = ref.flatten(), dev.flatten()
    assert set(fr) == set(fd) and all(same(fd[k], fr[k]) for k in fr)
    assert torch.equal(ref.flatten_packed()["legal_bits"], dev.flatten_packed()["legal_bits"])


def test_device_store_grows():
    steps = synth_steps(4, 400, seed=8, env_layout=False, overrides=True)     # 1600 rows > the initial 512 * 1
    host, dev = KataGoRolloutBuffer(1, (50, 9, 9), A), KataGoRolloutBuffer(1, (50, 9, 9), A)
    fill(host, steps, "cpu"); fill(dev, steps, DEV)
    fh, fd = host.flatten(), dev.flatten()
    assert all(same(fd[k], fh[k]) for k in fh)


def test_device_store_guards_raise_like_the_reference():
    N = 4
    ok = synth_steps(1, N, seed=1)[0]
    buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
    fill(buf, [ok], DEV)

    def broken(**kw):
        s = dict(ok)
        s.update(kw)
        return s

    term = torch.zeros(N, dtype=torch.bool); term[2] = True
    with pytest.raises(AssertionError, match="terminated must be a subset of dones"):
        fill(buf, [broken(dones=torch.zeros(N, dtype=torch.bool), terminated=term)], DEV)
    cats = ok["value_categories"].clone(); cats[1] = 7
    with pytest.raises(ValueError, match=r"invalid values \{7\}"):
        fill(buf, [broken(value_categories=cats)], DEV)
    sc = ok["score_targets"].clone(); sc[0] = float("nan")
    with pytest.raises(ValueError, match="contains NaN"):
        fill(buf, [broken(score_targets=sc)], DEV)
    sc = ok["score_targets"].clone(); sc[3] = -40.0
    with pytest.raises(ValueError, match="appear unnormalized: max abs value = 40.0"):
        fill(buf, [broken(score_targets=sc)], DEV)
    assert buf.size == 1 and buf.flatten()["rewards"].numel() == N       # rejected steps leave no rows behind
    fill(buf, [ok], DEV)
    assert buf.size == 2


def test_policy_loss_reads_packed_masks():
    B, S = 5, 9
    g = torch.Generator().manual_seed(12)
    logits = torch.randn(B, A, generator=g).to(DEV)
    legal = (torch.rand(S, A, generator=g) < 0.1)
    legal[:, 5] = True
    actions = torch.full((S,), 5, dtype=torch.long)
    idx = torch.tensor([8, 0, 3, 3, 6])
    bits = torch.empty(S, (A + 31) // 32, dtype=torch.int32, device=DEV)
    _lib.call("ka_pack_mask_bits", legal.to(DEV), bits, S, A, st())
    old, adv = (-8 + torch.randn(S, generator=g)).to(DEV), torch.randn(S, generator=g).to(DEV)
    outs = []
    for masks, words in ((legal.to(DEV), 0), (bits, (A + 31) // 32)):
        dl = torch.empty(B, A, device=DEV)
        nlp, rl, re = (torch.empty(B, device=DEV) for _ in range(3))
        flags = torch.zeros(2, dtype=torch.int32, device=DEV)
        _lib.call("ka_policy_loss", logits, masks, actions.to(DEV), old, adv, idx.to(DEV), dl, nlp, rl, re, flags, None,
                  0.2, 1.0 / B, 0.01 / B, B, A, words, st())
        outs.append((dl, nlp, rl, re, flags))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(_lib.KeiseiHipError, match="packed mask rows"):
        _lib.call("ka_policy_loss", logits, bits, actions.to(DEV), old, adv, idx.to(DEV), None, nlp, rl, re, flags, None,
                  0.2, 1.0 / B, 0.01 / B, B, A, 7, st())


def make_algo(golden):
    g = golden("g5_update")
    m = build_model("se_resnet", MP)
    m.load_state_dict(g.sub("sd0."))
    m.to(DEV)
    pp = KataGoPPOParams(learning_rate=1e-3, epochs_per_batch=2, batch_size=8, lambda_score=0.1, score_blend_alpha=0.1)
    return g, m, KataGoPPOAlgorithm(pp, m)


def test_update_from_device_store_matches_reference(golden, monkeypatch):
    """The reference's own update() result (golden g5) reproduced with every rollout tensor handed over on the device."""
    g, m, algo = make_algo(golden)
    T, N = 4, 4
    buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
    d = g.sub("buf.")
    for t in range(T):
        sl = slice(t * N, (t + 1) * N)
        buf.add(*[d[k][sl].to(DEV) for k in ("observations", "actions", "log_probs", "values", "rewards", "dones",
                                             "terminated", "legal_masks", "value_categories", "score_targets")],
                next_value_override=d["next_value_override"][sl].to(DEV))
    assert buf.is_device_resident
    it = iter(list(g["perms"]))
    monkeypatch.setattr(torch, "randperm", lambda n, *a, device=None, **k: next(it).to(device or "cpu"))
    copies = []
    real_pin = torch.Tensor.pin_memory
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: copies.append(self.shape) or real_pin(self, *a, **k))
    met = algo.update(buf, g["next_values"].to(DEV), value_adapter=MultiHeadValueAdapter(1.5, 0.1, 0.1))
    # nothing of the epoch (16 x 4050 observation floats, 16 x 11259 mask bytes) was staged through pinned host memory;
    # the pointer tables of the optimiser and of the grouped weight-gradient launch (<= 4096 words) are
    assert not [c for c in copies if c.numel() > 16384] and buf.size == 0
    for k in ("policy_loss", "value_loss", "score_loss", "entropy", "gradient_norm", "value_accuracy",
              "frac_predicted_win", "frac_predicted_draw", "frac_predicted_loss"):
        ref = float(g.np("metric." + k))
        assert abs(met[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, met[k], ref)
    ref_sd = g.sub("sd1.")
    got = m.state_dict()
    for k, v in ref_sd.items():
        if v.dtype.is_floating_point:       # same criterion as test_hip_ppo.test_fused_update_matches_reference
            diff = (got[k].cpu() - v).abs()
            assert float(diff.max()) <= 0.05 * 4e-3, (k, float(diff.max()))
            assert float((diff > 3e-5).float().mean()) <= 2e-3, k


@pytest.mark.parametrize("env_layout", [False, True])
def test_advantages_on_device_equal_host_path(golden, env_layout):
    """GAE for the grid and the per-environment (split-merge) layouts: the index arithmetic on the device builds
    the same padded grids as the host loop, so the scan results agree bit for bit."""
    _, m, algo = make_algo(golden)
    T, N = 9, 6
    steps = synth_steps(T, N, seed=21 + env_layout, env_layout=env_layout)
    host, dev = KataGoRolloutBuffer(N, (50, 9, 9), A), KataGoRolloutBuffer(N, (50, 9, 9), A)
    fill(host, steps, "cpu"); fill(dev, steps, DEV)
    nv = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    a_host = algo._advantages(host.flatten(), host, nv, torch.device(DEV))
    a_dev = algo._advantages(dev.flatten_packed(), dev, nv, torch.device(DEV))
    assert a_dev.is_cuda and not a_host.is_cuda and torch.equal(a_dev.cpu(), a_host)


def test_host_store_switch_and_pinned_batches(monkeypatch, tmp_path):
    """KA_ROLLOUT_BUFFER=host keeps the reference's CPU store although add() receives device tensors."""
    monkeypatch.setenv("KA_ROLLOUT_BUFFER", "host")
    steps = synth_steps(2, 3, seed=77)
    buf = KataGoRolloutBuffer(3, (50, 9, 9), A)
    fill(buf, steps, DEV)
    flat = buf.flatten()
    assert not buf.is_device_resident and not flat["observations"].is_cuda and flat["legal_masks"].dtype == torch.bool
    ref = KataGoRolloutBuffer(3, (50, 9, 9), A)
    fill(ref, steps, "cpu")
    assert all(same(flat[k], v) for k, v in ref.flatten().items())


def test_advantages_legacy_flat_layout_on_device(golden):
    """Rows that neither fill the (T, N) grid nor carry env ids: one chain, bootstrap from the mean next value."""
    _, m, algo = make_algo(golden)
    N = 4
    steps = synth_steps(5, N, seed=61, env_layout=True, overrides=False)
    for s in steps:
        s.pop("env_ids")
    host, dev = KataGoRolloutBuffer(N, (50, 9, 9), A), KataGoRolloutBuffer(N, (50, 9, 9), A)
    fill(host, steps, "cpu"); fill(dev, steps, DEV)
    if host.flatten()["rewards"].numel() == 5 * N:
        pytest.skip("the random step sizes happened to fill the grid")
    nv = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(DEV)
    a_host = algo._advantages(host.flatten(), host, nv, torch.device(DEV))
    a_dev = algo._advantages(dev.flatten_packed(), dev, nv, torch.device(DEV))
    assert torch.allclose(a_dev.cpu(), a_host.cpu(), rtol=1e-6, atol=1e-6)


def test_empty_step_raises_as_in_the_reference():
    s = synth_steps(1, 3, seed=5)[0]
    empty = {k: v[:0] for k, v in s.items()}
    for where in (DEV, "cpu"):
        buf = KataGoRolloutBuffer(3, (50, 9, 9), A)
        fill(buf, [s], where)
        with pytest.raises(RuntimeError, match="max"):
            fill(buf, [empty], where)
        assert buf.size == 1


def _sample(logits, masks, seed, vl=None, score=None, alpha=0.0):
    B, A_ = logits.shape
    acts = torch.empty(B, dtype=torch.int64, device=DEV); lp = torch.empty(B, device=DEV)
    vals = torch.empty(B, device=DEV) if vl is not None else None
    nl = torch.empty(B, dtype=torch.int32, device=DEV); flags = torch.zeros(2, dtype=torch.int32, device=DEV)
    _lib.call("ka_policy_sample", logits, int(logits.dtype == torch.bfloat16), masks, 0, seed, vl, score, alpha, acts, lp, vals,
              nl, flags, B, A_, st())
    return acts, lp, vals, nl, flags


def test_fused_sampler_draws_from_the_masked_softmax():
    """ka_policy_sample (the one-launch tail of select_actions, katago_ppo.py:567-612): legal actions only, log-probs of the
    masked softmax, the categorical distribution itself (frequencies over 40 000 draws), determinism in the seed, values."""
    g = torch.Generator(device=DEV).manual_seed(3)
    B, A_ = 6, 53
    logits = torch.randn(B, A_, device=DEV, generator=g) * 2
    masks = torch.rand(B, A_, device=DEV, generator=g) < 0.6
    masks[0] = False; masks[0, 17] = True                               # a single legal action
    masks[1] = True                                                     # everything legal
    masks[:, 5] |= ~masks.any(dim=1)
    ref = torch.log_softmax(logits.masked_fill(~masks, float("-inf")), dim=-1)
    counts = torch.zeros(B, A_, device=DEV)
    draws = 40000
    rows = torch.arange(B, device=DEV)
    for seed in range(draws // 50):
        big = logits.repeat(50, 1); bigm = masks.repeat(50, 1)          # the uniform depends on (seed, row): 300 distinct rows
        acts, lp, _, nl, flags = _sample(big, bigm, seed)
        assert bool(bigm[torch.arange(300, device=DEV), acts].all()) and flags.tolist() == [0, 0]
        assert torch.allclose(lp, ref.repeat(50, 1)[torch.arange(300, device=DEV), acts], atol=1e-5, rtol=1e-5)
        counts.index_put_((rows.repeat(50), acts), torch.ones(300, device=DEV), accumulate=True)
    assert torch.equal(nl[:B], masks.sum(1).int())
    p = ref.exp()
    sigma = (p * (1 - p) / draws).sqrt()
    assert bool(((counts / draws - p).abs() <= 4.5 * sigma + 1e-4).all()), ((counts / draws - p).abs() / (sigma + 1e-9)).max()
    assert counts[0, 17] == draws
    a1 = _sample(logits, masks, 12345)[0]; a2 = _sample(logits, masks, 12345)[0]
    assert torch.equal(a1, a2)
    assert any(not torch.equal(a1, _sample(logits, masks, s)[0]) for s in (1, 2, 3))
    # bf16 logits are read as such: same draws as their fp32 widening
    lb = logits.to(torch.bfloat16)
    ab, lpb = _sample(lb, masks, 99)[:2]; af, lpf = _sample(lb.float(), masks, 99)[:2]
    assert torch.equal(ab, af) and torch.equal(lpb, lpf)
    # values: P(W) - P(L), blended with the clamped score
    vl = torch.randn(B, 3, device=DEV, generator=g); sc = torch.randn(B, device=DEV, generator=g) * 2
    pr = torch.softmax(vl, -1)
    v0 = _sample(logits, masks, 1, vl)[2]
    assert torch.allclose(v0, pr[:, 0] - pr[:, 2], atol=1e-6)
    v1 = _sample(logits, masks, 1, vl, sc, 0.25)[2]
    assert torch.allclose(v1, 0.75 * (pr[:, 0] - pr[:, 2]) + 0.25 * sc.clamp(-1, 1), atol=1e-6)
    # a row without a legal action is flagged (select_actions raises on it)
    dead = masks.clone(); dead[3] = False
    assert _sample(logits, dead, 5)[4].tolist() == [0, 1]
    # full-size rows through select_actions: legal, log-probs of the model's masked softmax, reproducible under manual_seed
    torch.manual_seed(0)
    model = build_model("se_resnet", dict(MP)).to(DEV)
    algo = KataGoPPOAlgorithm(KataGoPPOParams(batch_size=64, use_amp=True), model)
    adapter = MultiHeadValueAdapter(1.5, 0.02, 0.3)
    obs = torch.randn(16, 50, 9, 9, device=DEV)
    legal = torch.rand(16, A, device=DEV) < 0.01; legal[:, 0] = True
    torch.manual_seed(7); r1 = algo.select_actions(obs, legal, adapter)
    torch.manual_seed(7); r2 = algo.select_actions(obs, legal, adapter)
    assert all(torch.equal(x, y) for x, y in zip(r1, r2))
    assert bool(legal[torch.arange(16, device=DEV), r1[0]].all())
    model.eval()
    with torch.no_grad():
        o = model(obs)
    model.train()
    want = torch.log_softmax(o.policy_logits.reshape(16, -1).float().masked_fill(~legal, float("-inf")), -1)
    assert torch.allclose(r1[1], want[torch.arange(16, device=DEV), r1[0]], atol=2e-3)
    assert torch.allclose(r1[2], adapter.scalar_value_blended(o.value_logits.float(), o.score_lead.float()), atol=2e-3)
    legal[9] = False
    with pytest.raises(RuntimeError, match=r"Environments \[9\] have zero legal actions"):
        algo.select_actions(obs, legal, adapter)
