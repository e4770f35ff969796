"""Rollout-side helpers of the reference's self-play loop (`split_merge_step` and what surrounds it).

Reference: keisei/training/katago_loop.py:63-431 -- `SplitMergeResult`, `_compute_value_cats`, `_negate_where`,
`to_learner_perspective`, `sign_correct_bootstrap`, `PendingTransitions`, `_resolve_opponent_devices`, `split_merge_step`.
Same names, arguments, results and error texts.  The `KataGoTrainingLoop` class itself (league, DB, checkpoints, display)
is the caller and stays the reference's; it can import these in place of its own.

What is different underneath: on a CUDA/HIP device nothing here goes through the host.  `current_players`,
`env_opponent_ids`, `learner_side` and the `condition` of `_negate_where` may be device tensors (the device VecEnv hands
them out, keisei_amd.shogi_gym) as well as the numpy arrays the reference passes; the learner / opponent partitions are
computed with device ops, masked softmax + sampling + log-prob are the one-launch HIP kernel `select_actions` uses
(`ka_policy_sample`),
and the forward passes run the eval-mode HIP path of the models.  The only synchronisation left is the reference's own
"zero legal actions" guard.  CPU tensors take the reference's tensor-op route (its tests run on the CPU).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F

from keisei_amd import _lib
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm

__all__ = ["SplitMergeResult", "_compute_value_cats", "_negate_where", "to_learner_perspective", "sign_correct_bootstrap",
           "PendingTransitions", "_resolve_opponent_devices", "split_merge_step"]


@dataclass
class SplitMergeResult:                                   # katago_loop.py:63-72
    actions: torch.Tensor            # (num_envs,) merged actions for all envs
    learner_mask: torch.Tensor       # (num_envs,) bool
    opponent_mask: torch.Tensor      # (num_envs,) bool
    learner_log_probs: torch.Tensor  # (n_learner,)
    learner_values: torch.Tensor     # (n_learner,) scalar values for GAE
    learner_indices: torch.Tensor    # (n_learner,) indices into the full env array


def _as_bool_tensor(condition: Any, device: torch.device) -> torch.Tensor:
    if isinstance(condition, torch.Tensor):
        return condition.to(device=device, dtype=torch.bool)
    return torch.from_numpy(np.ascontiguousarray(condition, dtype=np.bool_)).to(device)


def _differs(a: Any, b: Any) -> Any:
    """`a != b` for numpy arrays / ints / tensors in any mix (tensors win: the result stays on their device)."""
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        dev = a.device if isinstance(a, torch.Tensor) else b.device
        ta = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), device=dev)
        tb = b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b), device=dev)
        return ta.to(dev) != tb.to(dev)
    return a != b


def _compute_value_cats(rewards: torch.Tensor, terminal_mask: torch.Tensor, device: torch.device) -> torch.Tensor:
    """katago_loop.py:75-92: {-1 ignore, 0 win, 1 draw, 2 loss}; only genuinely terminal positions get a label."""
    cats = torch.full((rewards.numel(),), -1, dtype=torch.long, device=device)
    cats[terminal_mask & (rewards > 0)] = 0
    cats[terminal_mask & (rewards == 0)] = 1
    cats[terminal_mask & (rewards < 0)] = 2
    return cats


def _negate_where(values: torch.Tensor, condition: Any) -> torch.Tensor:
    """katago_loop.py:95-108: a copy with the elements where `condition` holds negated."""
    result = values.clone()
    if result.numel() == 0:
        return result
    mask = _as_bool_tensor(condition, values.device)
    return torch.where(mask, -result, result)


def to_learner_perspective(rewards: torch.Tensor, pre_players: Any, learner_side: Any) -> torch.Tensor:
    """katago_loop.py:111-122: rewards are the last mover's; flip them where the opponent moved."""
    return _negate_where(rewards, _differs(pre_players, learner_side))


def sign_correct_bootstrap(next_values: torch.Tensor, current_players: Any, learner_side: Any) -> torch.Tensor:
    """katago_loop.py:125-136: the value head speaks for the player to move; negate where that is the opponent."""
    return _negate_where(next_values, _differs(current_players, learner_side))


class PendingTransitions:
    """katago_loop.py:139-255: learner transitions waiting for their outcome (the opponent may move in between)."""

    def __init__(self, num_envs: int, obs_shape: tuple, action_space: int, device: torch.device) -> None:
        self.num_envs = num_envs
        self.obs = torch.zeros(num_envs, *obs_shape, device=device)
        self.actions = torch.zeros(num_envs, dtype=torch.long, device=device)
        self.log_probs = torch.zeros(num_envs, device=device)
        self.values = torch.zeros(num_envs, device=device)
        self.legal_masks = torch.zeros(num_envs, action_space, dtype=torch.bool, device=device)
        self.rewards = torch.zeros(num_envs, device=device)
        self.score_targets = torch.zeros(num_envs, device=device)
        self.valid = torch.zeros(num_envs, dtype=torch.bool, device=device)

    def create(self, env_mask: torch.Tensor, obs: torch.Tensor, actions: torch.Tensor, log_probs: torch.Tensor,
               values: torch.Tensor, legal_masks: torch.Tensor, rewards: torch.Tensor, score_targets: torch.Tensor) -> None:
        if (env_mask & self.valid).any():
            raise RuntimeError("create() called on env(s) with already-valid pending transition. "
                               "finalize() must be called first.")
        self.obs[env_mask] = obs[env_mask]
        self.actions[env_mask] = actions[env_mask]
        self.log_probs[env_mask] = log_probs[env_mask]
        self.values[env_mask] = values[env_mask]
        self.legal_masks[env_mask] = legal_masks[env_mask]
        self.rewards[env_mask] = rewards[env_mask]
        self.score_targets[env_mask] = score_targets[env_mask]
        self.valid[env_mask] = True

    def accumulate_reward(self, learner_rewards: torch.Tensor) -> None:
        self.rewards[self.valid] += learner_rewards[self.valid]

    def finalize(self, finalize_mask: torch.Tensor, dones: torch.Tensor, terminated: torch.Tensor):
        to_finalize = finalize_mask & self.valid
        if not to_finalize.any():
            return None
        indices = to_finalize.nonzero(as_tuple=True)[0]
        result = {
            "obs": self.obs[indices], "actions": self.actions[indices], "log_probs": self.log_probs[indices],
            "values": self.values[indices], "rewards": self.rewards[indices], "dones": dones[indices].float(),
            "terminated": terminated[indices].float(), "legal_masks": self.legal_masks[indices],
            "score_targets": self.score_targets[indices], "env_ids": indices,
        }
        self.valid[to_finalize] = False
        self.rewards[to_finalize] = 0.0
        return result


def _resolve_opponent_devices(opponents: dict, learner_device: torch.device) -> dict:
    """katago_loop.py:258-281: opponent id -> its device, or None when it shares the learner's."""
    if learner_device.type == "cuda" and learner_device.index is None:
        learner_device = torch.device(f"cuda:{torch.cuda.current_device()}")
    result: dict = {}
    for opp_id, model in opponents.items():
        try:
            opp_device = next(model.parameters()).device
        except (StopIteration, AttributeError):
            opp_device = learner_device
        result[opp_id] = opp_device if isinstance(opp_device, torch.device) and opp_device != learner_device else None
    return result


def _sample(logits: torch.Tensor, masks: torch.Tensor, who: str, env_index) -> tuple:
    """Masked softmax + one draw per row (katago_loop.py:345-356, 400-411).  Returns (actions, probs of the actions)."""
    rows, A = logits.shape
    if logits.is_cuda and masks.is_cuda and masks.dtype == torch.bool and logits.dtype in (torch.float32, torch.bfloat16):
        # masked softmax + one draw per row + log-prob in one launch (loss.hip ka_policy_sample, as select_actions)
        dev = logits.device
        lg = logits.contiguous()
        actions = torch.empty(rows, dtype=torch.int64, device=dev)
        log_probs = torch.empty(rows, device=dev)
        n_legal = torch.empty(rows, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        _lib.call("ka_policy_sample", lg, int(lg.dtype == torch.bfloat16), masks.contiguous(), 0, seed, None, None, 0.0, actions,
                  log_probs, None, n_legal, flags, rows, A, _lib.stream_ptr(dev))
        if int(flags[1].item()):
            raise RuntimeError(f"{who} envs {env_index(n_legal == 0)} have zero legal actions — all-False legal mask would produce NaN")
        return actions, log_probs
    n_legal = masks.sum(dim=-1)
    empty = n_legal == 0
    if bool(empty.any()):
        raise RuntimeError(f"{who} envs {env_index(empty)} have zero legal actions — all-False legal mask would produce NaN")
    probs = F.softmax(logits.masked_fill(~masks, float("-inf")), dim=-1)
    dist = torch.distributions.Categorical(probs, validate_args=False)
    actions = dist.sample()
    return actions, dist.log_prob(actions)


def split_merge_step(obs: torch.Tensor, legal_masks: torch.Tensor, current_players: Any, learner_model: torch.nn.Module,
                     opponent_model: torch.nn.Module | None = None, opponent_models: dict | None = None,
                     env_opponent_ids: Any = None, learner_side: Any = 0, value_adapter: Any | None = None,
                     opponent_devices: dict | None = None) -> SplitMergeResult:
    """katago_loop.py:284-431: one rollout step of league play -- the learner acts where it is to move, every opponent
    where it is, actions are merged; only learner-side log-probs / values come back."""
    if opponent_models is None and opponent_model is not None:
        active_opponents, active_env_ids = {0: opponent_model}, None
    elif opponent_models is not None:
        active_opponents, active_env_ids = opponent_models, env_opponent_ids
    else:
        raise ValueError("Must provide either opponent_model or opponent_models")

    num_envs, device = obs.shape[0], obs.device
    learner_mask = ~_as_bool_tensor(_differs(current_players, learner_side), device)
    opponent_mask = ~learner_mask
    learner_indices = learner_mask.nonzero(as_tuple=True)[0]

    actions = torch.zeros(num_envs, dtype=torch.long, device=device)
    learner_log_probs = torch.zeros(0, device=device)
    learner_values = torch.zeros(0, device=device)

    if learner_indices.numel() > 0:
        l_obs, l_masks = obs[learner_indices], legal_masks[learner_indices]
        learner_model.eval()                              # stays in eval: update() switches to train() itself
        with torch.no_grad():
            l_output = learner_model(l_obs)
        l_flat = l_output.policy_logits.reshape(l_obs.shape[0], -1)
        l_actions, learner_log_probs = _sample(l_flat, l_masks, "Learner", lambda e: learner_indices[e.to(learner_indices.device)].tolist())
        if value_adapter is not None:
            learner_values = value_adapter.scalar_value_blended(l_output.value_logits, l_output.score_lead)
        else:
            learner_values = KataGoPPOAlgorithm.scalar_value(l_output.value_logits)
        actions[learner_indices] = l_actions

    ids_t = None
    if active_env_ids is not None:
        ids_t = active_env_ids.to(device) if isinstance(active_env_ids, torch.Tensor) else torch.as_tensor(np.asarray(active_env_ids), device=device)
    for opp_id, model in active_opponents.items():
        opp_env_mask = opponent_mask if ids_t is None else (ids_t == opp_id) & opponent_mask
        idx_tensor = opp_env_mask.nonzero(as_tuple=True)[0]
        if idx_tensor.numel() == 0:
            continue
        o_obs, o_masks = obs[idx_tensor], legal_masks[idx_tensor]
        if opponent_devices is not None:
            opp_dev = opponent_devices.get(opp_id)
            cross_device = opp_dev is not None
        else:
            try:
                opp_dev = next(model.parameters()).device
            except (StopIteration, AttributeError):
                opp_dev = device
            cross_device = isinstance(opp_dev, torch.device) and opp_dev != device
        if cross_device:
            o_obs, o_masks = o_obs.to(opp_dev), o_masks.to(opp_dev)
        with torch.no_grad():
            o_output = model(o_obs)
        o_flat = o_output.policy_logits.reshape(o_obs.shape[0], -1)
        o_actions, _ = _sample(o_flat, o_masks, "Opponent", lambda e: idx_tensor[e.to(idx_tensor.device)].tolist())
        if cross_device:
            o_actions = o_actions.to(device)
        actions[idx_tensor] = o_actions

    return SplitMergeResult(actions=actions, learner_mask=learner_mask, opponent_mask=opponent_mask,
                            learner_log_probs=learner_log_probs, learner_values=learner_values, learner_indices=learner_indices)
