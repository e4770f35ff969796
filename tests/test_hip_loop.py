"""GPU: the calls of a self-play epoch in the order katago_loop.py makes them (select_actions -> buffer.add ... ->
fill_alternating_perspective_overrides -> update), repeated: everything stays on the device, the metrics stay finite and
in range, every minibatch took an optimiser step and device memory does not grow from epoch to epoch."""
import math

import pytest
import torch

from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm, KataGoPPOParams, KataGoRolloutBuffer
from keisei_amd.training.model_registry import build_model
from keisei_amd.training.value_adapter import MultiHeadValueAdapter

pytestmark = pytest.mark.gpu
DEV = "cuda"
A = 11259


def test_rollout_update_epochs():
    torch.manual_seed(0)
    model = build_model("se_resnet", dict(num_blocks=2, channels=64, se_reduction=8, global_pool_channels=32, policy_channels=16,
                                          value_fc_size=64, score_fc_size=32, obs_channels=50)).to(DEV)
    pp = KataGoPPOParams(learning_rate=5e-4, epochs_per_batch=2, batch_size=64, lambda_score=0.1, score_blend_alpha=0.1, use_amp=True)
    algo = KataGoPPOAlgorithm(pp, model)
    adapter = MultiHeadValueAdapter(pp.lambda_value, pp.lambda_score, pp.score_blend_alpha)
    T, N = 12, 16
    buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
    g = torch.Generator(device=DEV).manual_seed(1)
    legal = torch.zeros(N, A, dtype=torch.bool, device=DEV)
    legal[:, :500] = True
    good = torch.arange(N, device=DEV) % 500                     # the action each environment is rewarded for
    history, peak = [], []
    for epoch in range(4):
        obs = torch.randn(N, 50, 9, 9, device=DEV, generator=g)
        hits = 0.0
        for t in range(T):
            actions, logp, values = algo.select_actions(obs, legal, adapter)
            assert actions.is_cuda and bool(legal[torch.arange(N, device=DEV), actions].all())
            reward = (actions == good).float()
            hits += float(reward.mean())
            last = t == T - 1
            done = torch.full((N,), float(last), device=DEV)
            cats = torch.where(reward > 0, 0, 2) if last else torch.full((N,), -1, dtype=torch.long, device=DEV)
            buf.add(obs, actions, logp, values, reward, done, done, legal, cats, torch.zeros(N, device=DEV))
            assert buf.is_device_resident
        buf.fill_alternating_perspective_overrides()
        with torch.no_grad():
            model.eval()
            out = model(obs)
            model.train()
        met = algo.update(buf, adapter.scalar_value_blended(out.value_logits, out.score_lead), value_adapter=adapter)
        assert buf.size == 0 and all(math.isfinite(v) for v in met.values()), met
        history.append(met)
        torch.cuda.synchronize()
        peak.append(torch.cuda.memory_allocated())
    for met in history:                                                  # entropy over 500 legal actions: (0, ln 500]
        assert 0.0 < met["entropy"] <= math.log(500) + 1e-3 and met["gradient_norm"] > 0.0
    steps = float(next(iter(algo.optimizer.state.values()))["step"])
    assert steps == 4 * pp.epochs_per_batch * (T * N // pp.batch_size)   # 24 applied steps, none vetoed
    assert peak[-1] <= peak[1] * 1.02 + (1 << 20)                        # steady state: no per-epoch growth


def test_self_play_epochs_with_the_device_env():
    """SURVEY §8 f1 + f2 + f3 together: the games (keisei_amd.shogi_gym.VecEnv, torch output), the policy forward
    (select_actions), the rollout store and the PPO update all stay on the device -- no observation, mask or action
    crosses PCIe during an epoch."""
    from keisei_amd.shogi_gym import VecEnv

    torch.manual_seed(0)
    model = build_model("se_resnet", dict(num_blocks=2, channels=64, se_reduction=8, global_pool_channels=32, policy_channels=16,
                                          value_fc_size=64, score_fc_size=32, obs_channels=50)).to(DEV)
    pp = KataGoPPOParams(learning_rate=5e-4, epochs_per_batch=1, batch_size=128, lambda_score=0.1, score_blend_alpha=0.1, use_amp=True)
    algo = KataGoPPOAlgorithm(pp, model)
    adapter = MultiHeadValueAdapter(pp.lambda_value, pp.lambda_score, pp.score_blend_alpha)
    T, N = 16, 32
    env = VecEnv(num_envs=N, max_ply=24, observation_mode="katago", action_mode="spatial", output="torch", check_actions=False)
    buf = KataGoRolloutBuffer(N, (50, 9, 9), A)
    r = env.reset()
    obs, legal = r.observations, r.legal_masks
    for epoch in range(3):
        for t in range(T):
            actions, logp, values = algo.select_actions(obs, legal, adapter)
            res = env.step(actions)
            done = res.terminated | res.truncated
            cats = torch.where(done, torch.where(res.rewards > 0, 0, torch.where(res.rewards < 0, 2, 1)), -1)
            score = res.step_metadata.material_balance.float() / 76.0
            buf.add(obs, actions, logp, values, res.rewards, done.float(), res.terminated.float(), legal, cats, score)
            assert buf.is_device_resident
            obs, legal = res.observations, res.legal_masks
        env.raise_if_refused()                                           # every sampled action was legal
        buf.fill_alternating_perspective_overrides()
        with torch.no_grad():
            model.eval()
            out = model(obs)
            model.train()
        met = algo.update(buf, adapter.scalar_value_blended(out.value_logits, out.score_lead), value_adapter=adapter)
        assert buf.size == 0 and all(math.isfinite(v) for v in met.values()), met
    assert env.episodes_completed >= N                                   # max_ply 24 < 48 steps: every game ended at least once
    assert env.episodes_truncated > 0


def test_split_merge_step_with_device_players_and_the_device_env():
    """SURVEY §8 f2 (katago_loop.py:284-431): learner where it is to move, the opponents elsewhere, merged actions legal;
    players / opponent ids as device tensors (from the device env) and as the numpy arrays the reference passes."""
    import numpy as np

    from keisei_amd.shogi_gym import VecEnv
    from keisei_amd.training.katago_loop import (PendingTransitions, _compute_value_cats, sign_correct_bootstrap,
                                                  split_merge_step, to_learner_perspective)

    torch.manual_seed(0)
    cfg = dict(num_blocks=2, channels=64, se_reduction=8, global_pool_channels=32, policy_channels=16, value_fc_size=64,
               score_fc_size=32, obs_channels=50)
    learner, opp_a, opp_b = (build_model("se_resnet", dict(cfg)).to(DEV).eval() for _ in range(3))
    N = 48
    env = VecEnv(num_envs=N, max_ply=40, observation_mode="katago", action_mode="spatial", output="torch", check_actions=False)
    r = env.reset()
    obs, legal = r.observations, r.legal_masks
    players = torch.zeros(N, dtype=torch.uint8, device=DEV)
    opp_ids = torch.arange(N, device=DEV) % 2
    learner_side = (torch.arange(N, device=DEV) % 3 == 0).to(torch.uint8)       # per-env sides, as the league assigns them
    pending = PendingTransitions(N, (50, 9, 9), A, torch.device(DEV))
    seen_learner = seen_opp = 0
    for t in range(30):
        sm = split_merge_step(obs, legal, players, learner, opponent_models={0: opp_a, 1: opp_b}, env_opponent_ids=opp_ids,
                              learner_side=learner_side)
        assert torch.equal(sm.learner_mask, players == learner_side) and torch.equal(sm.opponent_mask, ~sm.learner_mask)
        assert torch.equal(sm.learner_indices, sm.learner_mask.nonzero(as_tuple=True)[0])
        assert bool(legal[torch.arange(N, device=DEV), sm.actions].all())
        n_l = int(sm.learner_mask.sum())
        assert sm.learner_log_probs.shape == (n_l,) and sm.learner_values.shape == (n_l,)
        assert bool(torch.isfinite(sm.learner_log_probs).all()) and bool((sm.learner_log_probs <= 0).all())
        if n_l:                                                          # log-probs are the masked softmax of the learner's logits
            with torch.no_grad():
                logits = learner(obs[sm.learner_indices]).policy_logits.reshape(n_l, -1).float()
            ref = torch.log_softmax(logits.masked_fill(~legal[sm.learner_indices], float("-inf")), dim=-1)
            got = ref.gather(1, sm.actions[sm.learner_indices].unsqueeze(1)).squeeze(1)
            assert torch.allclose(sm.learner_log_probs, got, atol=2e-3, rtol=0)
        seen_learner += n_l; seen_opp += N - n_l
        pre_players, pre_copy = players, players.clone()
        res = env.step(sm.actions)
        players = res.current_players
        # a result outlives the next step (every per-step field alternates between two buffers): the players handed out
        # before this step are still the side that just moved -- in a running game the other side is now to move, a
        # finished game was set up again with black to move (the reference keeps `current_players.copy()`, katago_loop.py)
        assert torch.equal(pre_players, pre_copy) and players.data_ptr() != pre_players.data_ptr()
        over = res.terminated | res.truncated
        assert torch.equal(players[~over], 1 - pre_players[~over]) and bool((players[over] == 0).all())
        rew = to_learner_perspective(res.rewards, pre_players, learner_side)
        assert torch.equal(rew.abs(), res.rewards.abs())
        flipped = pre_players != learner_side
        assert torch.equal(rew[flipped], -res.rewards[flipped]) and torch.equal(rew[~flipped], res.rewards[~flipped])
        boot = sign_correct_bootstrap(torch.ones(N, device=DEV), players, learner_side)
        assert torch.equal(boot == -1, players != learner_side)
        cats = _compute_value_cats(rew, res.terminated, torch.device(DEV))
        assert bool((cats[~res.terminated] == -1).all())
        pending.accumulate_reward(rew)
        done = res.terminated | res.truncated
        fin = pending.finalize(done | sm.learner_mask, done, res.terminated)
        if fin is not None:
            assert fin["obs"].shape[1:] == (50, 9, 9) and fin["env_ids"].dtype == torch.long
        full_lp = torch.zeros(N, device=DEV); full_lp[sm.learner_indices] = sm.learner_log_probs
        full_v = torch.zeros(N, device=DEV); full_v[sm.learner_indices] = sm.learner_values
        pending.create(sm.learner_mask & ~done, obs, sm.actions, full_lp, full_v, legal, torch.zeros(N, device=DEV),
                       torch.zeros(N, device=DEV))
        obs, legal = res.observations, res.legal_masks
    env.raise_if_refused()
    assert seen_learner > 0 and seen_opp > 0
    # the reference's call form: numpy players / ids, one opponent model, integer side
    sm = split_merge_step(obs, legal, players.cpu().numpy(), learner, opponent_model=opp_a, learner_side=0)
    assert torch.equal(sm.learner_mask, players == 0) and bool(legal[torch.arange(N, device=DEV), sm.actions].all())
    with pytest.raises(ValueError, match="Must provide either opponent_model or opponent_models"):
        split_merge_step(obs, legal, players, learner)
    dead = legal.clone(); dead[5] = False
    with pytest.raises(RuntimeError, match="zero legal actions"):
        split_merge_step(obs, dead, torch.full((N,), 0, dtype=torch.uint8, device=DEV), learner, opponent_model=opp_a, learner_side=0)


def test_pending_transitions_device_store_equals_the_host_store():
    """The slot store of the split-merge protocol (keisei_amd/training/katago_loop.py; reference katago_loop.py:139-250): the
    device backend (one `ka_pending_open` / `ka_pending_settle` launch per call, packed masks) against the host backend on the
    same random protocol -- every settled row, the value labels, the slots left open; masks handed over as bool rows and as
    the device env's packed rows; the fused reward accumulation; the "slot still taken" guard."""
    import numpy as np
    from keisei_amd.training.katago_loop import PendingTransitions, _compute_value_cats

    N, shape, Asp = 37, (50, 9, 9), A
    dev_store = PendingTransitions(N, shape, Asp, torch.device(DEV))
    cpu_store = PendingTransitions(N, shape, Asp, torch.device("cpu"))
    assert dev_store.legal_mask_bits.shape == (N, 352) and not hasattr(dev_store, "_legal_masks")
    rng = np.random.default_rng(3)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    settled_rows = 0
    for step in range(40):
        lr = t(rng.integers(-1, 2, N).astype(np.float32))               # this step's learner-perspective rewards
        dones = t(rng.random(N) < 0.2); terminated = dones & t(rng.random(N) < 0.7)
        fin = t(rng.random(N) < 0.5) | dones
        fuse = step % 2 == 0                                            # accumulate_reward() as its own call / inside finalize()
        if not fuse:
            dev_store.accumulate_reward(lr.to(DEV)); cpu_store.accumulate_reward(lr)
        as_f = step % 3 != 0                                            # the loop passes float flags; bool flags work too
        flag = (lambda x: x.float()) if as_f else (lambda x: x)
        got = dev_store.finalize(fin.to(DEV), flag(dones).to(DEV), flag(terminated).to(DEV), accumulate=lr.to(DEV) if fuse else None)
        want = cpu_store.finalize(fin, flag(dones), flag(terminated), accumulate=lr if fuse else None)
        assert (got is None) == (want is None)
        if want is not None:
            settled_rows += want["env_ids"].numel()
            for k in ("obs", "actions", "log_probs", "values", "rewards", "dones", "terminated", "score_targets", "env_ids",
                      "legal_mask_bits", "value_cats", "legal_masks"):
                assert torch.equal(got[k].cpu(), want[k]), (step, k)
            assert torch.equal(got["value_cats"].cpu(), _compute_value_cats(want["rewards"], want["terminated"].bool(), torch.device("cpu")))
            assert got["legal_masks"].dtype == torch.bool and got["legal_masks"].shape == (want["env_ids"].numel(), Asp)
        assert torch.equal(dev_store.valid.cpu(), cpu_store.valid) and torch.equal(dev_store.rewards.cpu(), cpu_store.rewards)
        # open new slots where none is pending
        env_mask = t(rng.random(N) < 0.6) & ~cpu_store.valid
        obs = t(rng.standard_normal((N, *shape)).astype(np.float32))
        masks = t(rng.random((N, Asp)) < 0.05)
        acts = t(rng.integers(0, Asp, N)); lp = t(-rng.random(N).astype(np.float32)); val = t(rng.standard_normal(N).astype(np.float32))
        rew = t(rng.integers(-1, 2, N).astype(np.float32)); sc = t(rng.standard_normal(N).astype(np.float32))
        cpu_store.create(env_mask, obs, acts, lp, val, masks, rew, sc)
        if step % 2:                                                    # packed rows, as the device env hands them out
            bits = torch.empty(N, 352, dtype=torch.int32, device=DEV)
            from keisei_amd import _lib
            _lib.call("ka_pack_mask_bits", masks.to(DEV), bits, N, Asp, _lib.stream_ptr())
            dev_masks = bits
        else:
            dev_masks = masks.to(DEV)
        dev_store.create(env_mask.to(DEV), obs.to(DEV), acts.to(DEV), lp.to(DEV), val.to(DEV), dev_masks, rew.to(DEV), sc.to(DEV))
        for name in ("obs", "actions", "log_probs", "values", "rewards", "score_targets", "legal_mask_bits", "valid"):
            assert torch.equal(getattr(dev_store, name).cpu(), getattr(cpu_store, name)), (step, name)
    assert settled_rows > 100
    assert torch.equal(dev_store.legal_masks.cpu(), cpu_store.legal_masks)
    # a slot that is still open must not be opened again: nothing is written, RuntimeError like the reference's
    taken = cpu_store.valid.clone()
    assert bool(taken.any())
    before = {k: getattr(dev_store, k).clone() for k in ("obs", "actions", "rewards", "legal_mask_bits", "valid")}
    args = (torch.zeros(N, *shape), torch.zeros(N, dtype=torch.long), torch.zeros(N), torch.zeros(N),
            torch.ones(N, Asp, dtype=torch.bool), torch.zeros(N), torch.zeros(N))
    with pytest.raises(RuntimeError, match="already-valid pending transition"):
        dev_store.create(torch.ones(N, dtype=torch.bool, device=DEV), *[a.to(DEV) for a in args])
    with pytest.raises(RuntimeError, match="already-valid pending transition"):
        cpu_store.create(torch.ones(N, dtype=torch.bool), *args)
    for k, v in before.items():
        assert torch.equal(getattr(dev_store, k), v), k
