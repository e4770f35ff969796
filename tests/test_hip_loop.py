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
