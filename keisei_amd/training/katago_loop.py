"""Rollout-side helpers of the reference's self-play loop: the split learner / opponent step and the
pending-transition protocol around it.

Boundary kept from keisei/training/katago_loop.py:63-431: the names `SplitMergeResult`, `_compute_value_cats`,
`_negate_where`, `to_learner_perspective`, `sign_correct_bootstrap`, `PendingTransitions`, `_resolve_opponent_devices`,
`split_merge_step`, their arguments, results and error texts (the reference's own split-merge / loop tests run against this
file through tools/conformance.sh --loop-helpers).  The `KataGoTrainingLoop` class is the caller and stays the reference's.

Built for the device, not re-typed from the reference:

* `PendingTransitions` is a slot store in HBM -- one row per game: observation, the legal mask as a PACKED row (352 words
  instead of 11 259 bytes: what the device env and the device rollout store already speak), scalars, a valid byte.  Opening
  slots (`create`) is one launch of `ka_pending_open` (scatter by game mask, masks packed or copied on the way, the "slot
  still taken" guard as a device flag); settling them (`finalize`) is one launch of `ka_pending_settle` (optional reward
  accumulation, selection, stable compaction in game order, value-head labels, slot release) -- where the reference moves
  eight tensors with boolean-mask indexing and a synchronising nonzero().  Bool masks are rebuilt only for a caller that
  asks for them (`result["legal_masks"]`, `pending.legal_masks`).
* `split_merge_step` seats every game once (learner, or the id of the opponent to move), sorts the games by seat and reads the
  group sizes back in ONE transfer; each model then runs on a contiguous slice, with masked softmax + draw + log-prob in the
  one-launch HIP kernel `select_actions` uses (`ka_policy_sample`).  `current_players`, `env_opponent_ids` and `learner_side`
  may be device tensors (the device VecEnv hands them out) as well as numpy arrays.
* CPU tensors (the reference's tests are CPU tests) take a host backend with the same packed layout (numpy bit packing, index
  gathers); it is device dispatch, not a fallback for the GPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import numpy as np
import torch
import torch.nn.functional as F

from keisei_amd import _lib
from keisei_amd.training.katago_ppo import KataGoPPOAlgorithm

__all__ = ["SplitMergeResult", "_compute_value_cats", "_negate_where", "to_learner_perspective", "sign_correct_bootstrap",
           "PendingTransitions", "_resolve_opponent_devices", "split_merge_step"]

_ZERO_LEGAL = "{who} envs {envs} have zero legal actions — all-False legal mask would produce NaN"


@dataclass
class SplitMergeResult:
    """What one split-merge step hands back (katago_loop.py:63-72): merged actions for every game, and the learner's side only
    of log-probs / values (the rollout buffer stores learner transitions only)."""
    actions: torch.Tensor            # (num_envs,)
    learner_mask: torch.Tensor       # (num_envs,) bool
    opponent_mask: torch.Tensor      # (num_envs,) bool
    learner_log_probs: torch.Tensor  # (n_learner,)
    learner_values: torch.Tensor     # (n_learner,)
    learner_indices: torch.Tensor    # (n_learner,) game indices, ascending


# ---------------------------------------------------------------------------------------------- small tensor helpers
def _on(device: torch.device, x: Any, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """numpy array / scalar / tensor -> tensor on `device` (tensors already there are used in place)."""
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=device, dtype=dtype) if (t.device != device or (dtype is not None and t.dtype != dtype)) else t


def _differs(a: Any, b: Any, device: torch.device) -> torch.Tensor:
    """Element-wise a != b as a bool tensor on `device`, for any mix of numpy arrays, ints and tensors."""
    if not isinstance(a, torch.Tensor) and not isinstance(b, torch.Tensor):
        return _on(device, np.ascontiguousarray(np.asarray(a) != np.asarray(b)), torch.bool)
    return _on(device, a) != _on(device, b)


def _compute_value_cats(rewards: torch.Tensor, terminal_mask: torch.Tensor, device: torch.device) -> torch.Tensor:
    """Value-head labels {-1 ignore, 0 win, 1 draw, 2 loss} (katago_loop.py:75-92): a terminal position is labelled by the sign
    of its (integer-valued) reward, everything else is ignored.  `finalize()` returns the same labels as "value_cats"."""
    by_sign = (1 - torch.sign(rewards.reshape(-1))).to(device=device, dtype=torch.long)       # +1 -> 0, 0 -> 1, -1 -> 2
    return torch.where(terminal_mask.reshape(-1).to(device), by_sign, torch.full_like(by_sign, -1))


def _negate_where(values: torch.Tensor, condition: Any) -> torch.Tensor:
    """A fresh tensor equal to `values` with the sign flipped where `condition` holds (katago_loop.py:95-108)."""
    if values.numel() == 0:
        return values.clone()
    return torch.where(_on(values.device, condition, torch.bool), -values, values)


def to_learner_perspective(rewards: torch.Tensor, pre_players: Any, learner_side: Any) -> torch.Tensor:
    """Step rewards speak for the side that just moved (`pre_players`); the learner's view flips them where that was the
    opponent (katago_loop.py:111-122)."""
    return _negate_where(rewards, _differs(pre_players, learner_side, rewards.device))


def sign_correct_bootstrap(next_values: torch.Tensor, current_players: Any, learner_side: Any) -> torch.Tensor:
    """The value head speaks for the side to move; learner-centred GAE needs it negated where that is the opponent
    (katago_loop.py:125-136)."""
    return _negate_where(next_values, _differs(current_players, learner_side, next_values.device))


# ---------------------------------------------------------------------------------------------- pending transitions
class _Settled(dict):
    """finalize()'s result: the reference's keys, plus "legal_mask_bits" (packed rows, what the device rollout store takes
    as they are) and "value_cats".  "legal_masks" (bool rows) is rebuilt from the packed rows on first access."""

    def __init__(self, owner: "PendingTransitions", **kw):
        super().__init__(**kw)
        self._owner = owner

    def __missing__(self, key):
        if key != "legal_masks":
            raise KeyError(key)
        masks = self._owner._unpack(self["legal_mask_bits"])
        self[key] = masks
        return masks


class PendingTransitions:
    """Learner transitions waiting for their outcome (katago_loop.py:139-250): opened when the learner moves, settled when
    the game ends or the turn comes back -- the opponent's reply in between still adds to the reward.

    One slot per game.  Columns: `obs` (num_envs, *obs_shape) fp32, `legal_mask_bits` (num_envs, ceil(A/32)) int32 packed rows,
    `actions` int64, `log_probs` / `values` / `rewards` / `score_targets` fp32, `valid` bool.  At 512 games: 33 MB of
    observations + 0.7 MB of masks (the reference's bool mask buffer alone is 5.8 MB).  `legal_masks` is a property that
    unpacks on demand; nothing keeps bool rows."""

    def __init__(self, num_envs: int, obs_shape: tuple, action_space: int, device: torch.device, *, check: bool = True) -> None:
        device = torch.device(device)
        self.num_envs, self.action_space, self.device = int(num_envs), int(action_space), device
        self._words = (self.action_space + 31) // 32
        self._obs_elems = int(np.prod(obs_shape)) if len(obs_shape) else 1
        self._gpu = device.type == "cuda"
        self._check = check                       # False: the "slot still taken" guard is read by raise_if_conflict() instead
        if self._gpu:
            _lib._load()                          # the device backend is the HIP library; it raises when that is missing
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=device)
        self.obs = z(num_envs, *obs_shape)
        self.legal_mask_bits = z(num_envs, self._words, dtype=torch.int32)
        self.actions = z(num_envs, dtype=torch.long)
        self.log_probs, self.values, self.rewards, self.score_targets = z(num_envs), z(num_envs), z(num_envs), z(num_envs)
        self._valid = [z(num_envs, dtype=torch.bool), z(num_envs, dtype=torch.bool)]      # settle writes the other one
        self._cur = 0
        self._flags = z(2, dtype=torch.int32)

    # ---- views
    @property
    def valid(self) -> torch.Tensor:
        return self._valid[self._cur]

    @property
    def legal_masks(self) -> torch.Tensor:
        """(num_envs, action_space) bool, rebuilt from the packed rows (a copy: writing to it changes nothing)."""
        return self._unpack(self.legal_mask_bits)

    def _unpack(self, bits: torch.Tensor) -> torch.Tensor:
        rows = bits.shape[0]
        out = torch.empty(rows, self.action_space, dtype=torch.bool, device=bits.device)
        if rows == 0:
            return out
        if bits.is_cuda:
            _lib.call("ka_unpack_mask_bits", bits.contiguous(), None, out, rows, self.action_space, _lib.stream_ptr(bits.device))
            return out
        words = np.ascontiguousarray(bits.numpy()).view(np.uint8)                       # little-endian words -> bytes
        return torch.from_numpy(np.unpackbits(words, axis=1, bitorder="little")[:, :self.action_space].astype(np.bool_))

    def _pack_host(self, masks: torch.Tensor) -> torch.Tensor:
        m = np.ascontiguousarray(masks.numpy().astype(np.uint8))
        packed = np.zeros((m.shape[0], self._words * 4), np.uint8)
        bits = np.packbits(m, axis=1, bitorder="little")
        packed[:, :bits.shape[1]] = bits
        return torch.from_numpy(packed.view(np.int32))

    def _mask_rows(self, legal_masks: torch.Tensor):
        """(bool rows or None, packed rows or None) of what create() was handed."""
        if legal_masks.dtype == torch.int32 and legal_masks.shape[-1] == self._words:
            return None, legal_masks
        return legal_masks, None

    # ---- protocol
    def create(self, env_mask: torch.Tensor, obs: torch.Tensor, actions: torch.Tensor, log_probs: torch.Tensor,
               values: torch.Tensor, legal_masks: torch.Tensor, rewards: torch.Tensor, score_targets: torch.Tensor) -> None:
        """Open the slots of the games in `env_mask` with this step's rows (all arguments carry num_envs rows).  `legal_masks`:
        (num_envs, A) bool, or the packed (num_envs, ceil(A/32)) int32 rows of the device env.  A game that still holds a
        pending transition is a protocol error: nothing is written and RuntimeError is raised (finalize() comes first)."""
        bool_rows, packed_rows = self._mask_rows(legal_masks)
        if self._gpu:
            dev = self.device
            c = lambda t, dt: t.to(device=dev, dtype=dt).contiguous()
            _lib.call("ka_pending_open", self.obs, self.legal_mask_bits, self.actions, self.log_probs, self.values, self.rewards,
                      self.score_targets, self.valid, c(env_mask, torch.bool), c(obs, torch.float32),
                      None if bool_rows is None else c(bool_rows, torch.bool), None if packed_rows is None else c(packed_rows, torch.int32),
                      c(actions, torch.long), c(log_probs, torch.float32), c(values, torch.float32), c(rewards, torch.float32),
                      c(score_targets, torch.float32), self._flags, self.num_envs, self._obs_elems, self.action_space,
                      _lib.stream_ptr(dev))
            if self._check:
                self.raise_if_conflict()
            return
        games = torch.nonzero(env_mask, as_tuple=True)[0]
        if bool(self.valid[games].any()):
            self._conflict()
        for column, rows in ((self.obs, obs), (self.actions, actions), (self.log_probs, log_probs), (self.values, values),
                             (self.rewards, rewards), (self.score_targets, score_targets)):
            column.index_copy_(0, games, rows.index_select(0, games).to(column.dtype))
        picked = legal_masks.index_select(0, games)
        self.legal_mask_bits.index_copy_(0, games, picked if packed_rows is not None else self._pack_host(picked))
        self.valid.index_fill_(0, games, True)

    @staticmethod
    def _conflict():
        raise RuntimeError("create() called on env(s) with already-valid pending transition. "
                           "finalize() must be called first.")

    def raise_if_conflict(self) -> None:
        """Device backend: reads the guard flag of the last create() (one 4-byte transfer)."""
        if self._gpu and int(self._flags[0].item()):
            self._conflict()

    def accumulate_reward(self, learner_rewards: torch.Tensor) -> None:
        """Add this step's learner-perspective rewards to every open slot (katago_loop.py:203-211; non-movers carry 0)."""
        if self._gpu:
            _lib.call("ka_pending_accumulate", self.rewards, self.valid,
                      learner_rewards.to(device=self.device, dtype=torch.float32).contiguous(), self.num_envs,
                      _lib.stream_ptr(self.device))
            return
        self.rewards.copy_(torch.where(self.valid, self.rewards + learner_rewards.to(self.rewards.dtype), self.rewards))

    def finalize(self, finalize_mask: torch.Tensor, dones: torch.Tensor, terminated: torch.Tensor, *,
                 accumulate: Optional[torch.Tensor] = None):
        """Settle the open slots selected by `finalize_mask` (games without an open slot are skipped).  Returns None when
        there is nothing to settle, else a dict with rows in game order: obs, actions, log_probs, values, rewards, dones,
        terminated (floats), legal_masks, score_targets, env_ids -- plus legal_mask_bits and value_cats.  `accumulate`: this
        step's learner rewards, added first (accumulate_reward() folded into the same launch).
        On the device the rows are views into two persistent column sets used in turn: a settled batch stays valid until the
        finalize() after the next one (the loop hands it to the rollout store at once, as the reference does, katago_loop.py:1523)."""
        if self._gpu:
            return self._settle_device(finalize_mask, dones, terminated, accumulate)
        if accumulate is not None:
            self.accumulate_reward(accumulate)
        games = torch.nonzero(finalize_mask & self.valid, as_tuple=True)[0]
        if games.numel() == 0:
            return None
        take = lambda t: t.index_select(0, games)
        rewards = take(self.rewards)
        term = take(terminated).float()
        out = _Settled(self, obs=take(self.obs), actions=take(self.actions), log_probs=take(self.log_probs), values=take(self.values),
                       rewards=rewards, dones=take(dones).float(), terminated=term, score_targets=take(self.score_targets),
                       env_ids=games, legal_mask_bits=take(self.legal_mask_bits),
                       value_cats=_compute_value_cats(rewards, term.bool(), self.device))
        self.valid.index_fill_(0, games, False)
        self.rewards.index_fill_(0, games, 0.0)
        return out

    def _settle_device(self, finalize_mask, dones, terminated, accumulate):
        dev, n = self.device, self.num_envs
        flags_f32 = int(dones.dtype == torch.float32 and terminated.dtype == torch.float32)
        as_flag = (lambda t: t.to(device=dev, dtype=torch.float32).contiguous()) if flags_f32 else \
                  (lambda t: (t != 0).to(device=dev).contiguous())
        # the output columns are two persistent sets used in turn (no num_envs-row allocations per call, and the [:k]
        # views handed out do not pin fresh full-size blocks): a settled batch stays valid until the finalize AFTER the next one
        sets = getattr(self, "_out_sets", None)
        if sets is None:
            e = lambda *s, dtype=torch.float32: torch.empty(*s, dtype=dtype, device=dev)
            sets = self._out_sets = [dict(obs=e(n, *self.obs.shape[1:]), legal_mask_bits=e(n, self._words, dtype=torch.int32),
                                          actions=e(n, dtype=torch.long), log_probs=e(n), values=e(n), rewards=e(n), dones=e(n),
                                          terminated=e(n), score_targets=e(n), env_ids=e(n, dtype=torch.long),
                                          value_cats=e(n, dtype=torch.long)) for _ in range(2)]
            self._out_flip = 0
        o = sets[self._out_flip]
        self._out_flip ^= 1
        nxt = self._cur ^ 1
        _lib.call("ka_pending_settle", self.obs, self.legal_mask_bits, self.actions, self.log_probs, self.values, self.rewards,
                  self.score_targets, self._valid[self._cur], self._valid[nxt],
                  finalize_mask.to(device=dev, dtype=torch.bool).contiguous(), as_flag(dones), as_flag(terminated), flags_f32,
                  None if accumulate is None else accumulate.to(device=dev, dtype=torch.float32).contiguous(),
                  o["obs"], o["legal_mask_bits"], o["actions"], o["log_probs"], o["values"], o["rewards"], o["dones"],
                  o["terminated"], o["score_targets"], o["env_ids"], o["value_cats"], self._flags, n, self._obs_elems,
                  self.action_space, _lib.stream_ptr(dev))
        self._cur = nxt
        k = int(self._flags[1].item())            # the one transfer of the call: how many rows came out (the reference's nonzero())
        if k == 0:
            return None
        return _Settled(self, **{name: t[:k] for name, t in o.items()})


# ---------------------------------------------------------------------------------------------- split-merge step
def _device_of(model: Any, default: torch.device) -> torch.device:
    params = getattr(model, "parameters", None)
    first = next(iter(params()), None) if callable(params) else None
    return first.device if isinstance(first, torch.Tensor) else default


def _resolve_opponent_devices(opponents: dict, learner_device: torch.device) -> dict:
    """Opponent id -> the device its model lives on, or None where that is the learner's (no transfer needed); computed once
    per epoch instead of probing every model on every step (katago_loop.py:253-281)."""
    home = learner_device
    if home.type == "cuda" and home.index is None:
        home = torch.device("cuda", torch.cuda.current_device())
    where = {opp_id: _device_of(model, home) for opp_id, model in opponents.items()}
    return {opp_id: (dev if dev != home else None) for opp_id, dev in where.items()}


def _draw(logits: torch.Tensor, masks: torch.Tensor, who: str, games: torch.Tensor):
    """One action per row from the masked softmax, and its log-probability.  `games`: the game index of every row (for the
    zero-legal-actions error).  Device rows take the one-launch kernel of select_actions; bool or packed int32 masks."""
    rows, A = logits.shape
    packed = masks.dtype == torch.int32
    if logits.is_cuda and masks.is_cuda and logits.dtype in (torch.float32, torch.bfloat16):
        dev = logits.device
        lg = logits.contiguous()
        actions = torch.empty(rows, dtype=torch.int64, device=dev)
        log_probs = torch.empty(rows, device=dev)
        n_legal = torch.empty(rows, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())      # host generator: torch.manual_seed() fixes the rollout
        _lib.call("ka_policy_sample", lg, int(lg.dtype == torch.bfloat16), masks.contiguous(), masks.shape[1] if packed else 0, seed,
                  None, None, 0.0, actions, log_probs, None, n_legal, flags, rows, A, _lib.stream_ptr(dev))
        nan_seen, zero_legal = flags.tolist()                                     # the one synchronisation of the draw
        if zero_legal:
            raise RuntimeError(_ZERO_LEGAL.format(who=who, envs=games[(n_legal == 0).to(games.device)].tolist()))
        if nan_seen:
            raise RuntimeError(f"NaN in raw policy logits of the {who.lower()} — probability tensor contains nan")
        return actions, log_probs
    if packed:
        raise TypeError("packed legal masks need device tensors (CPU rows carry bool masks)")
    playable = masks.any(dim=-1)
    if not bool(playable.all()):
        raise RuntimeError(_ZERO_LEGAL.format(who=who, envs=games[(~playable).to(games.device)].tolist()))
    dist = torch.distributions.Categorical(F.softmax(logits.masked_fill(~masks, float("-inf")), dim=-1), validate_args=False)
    actions = dist.sample()
    return actions, dist.log_prob(actions)


def split_merge_step(obs: torch.Tensor, legal_masks: torch.Tensor, current_players: Any, learner_model: torch.nn.Module,
                     opponent_model: torch.nn.Module | None = None, opponent_models: dict | None = None,
                     env_opponent_ids: Any = None, learner_side: Any = 0, value_adapter: Any | None = None,
                     opponent_devices: dict | None = None) -> SplitMergeResult:
    """One rollout step of league play (katago_loop.py:284-431): the learner acts in the games where it is to move, each
    opponent in its own games where it is to move; the actions are merged, and only the learner's log-probs / values are
    returned.  Single-opponent form: `opponent_model`; cohort form: `opponent_models` {id: model} + `env_opponent_ids`."""
    if opponent_models is not None:
        cohort, game_opponent = opponent_models, env_opponent_ids
    elif opponent_model is not None:
        cohort, game_opponent = {0: opponent_model}, None
    else:
        raise ValueError("Must provide either opponent_model or opponent_models")

    n, dev = obs.shape[0], obs.device
    opponent_mask = _differs(current_players, learner_side, dev)
    learner_mask = ~opponent_mask
    # seat of every game: -1 = the learner moves, k >= 0 = opponent k moves.  A stable sort by seat makes every seat's games a
    # contiguous, ascending run; the run lengths are the only thing read back (one transfer for the whole step).
    ranked = sorted(cohort)
    rank_of = {opp_id: r for r, opp_id in enumerate(ranked)}
    if game_opponent is None:
        # no per-game assignment: the reference lets EVERY model act on all opponent games, in dict order, and keeps the last one's
        # actions (katago_loop.py:404-431).  Here only that last model runs -- the merged actions are its own either way -- and each
        # skipped model still consumes the one host draw its sampling would have taken (below), so a seeded rollout stays in
        # step with the all-models form.  Deviation: the zero-legal / NaN guards of the skipped models do not fire.
        slot_of = torch.full((n,), rank_of[list(cohort)[-1]], dtype=torch.long, device=dev)
    else:
        wanted = _on(dev, game_opponent, torch.long)
        table = torch.tensor(ranked, dtype=torch.long, device=dev)
        slot_of = torch.bucketize(wanted, table).clamp_(max=len(ranked) - 1)
        slot_of = torch.where(table[slot_of] == wanted, slot_of, torch.full_like(slot_of, len(ranked)))   # unknown id: unseated
    seat = torch.where(learner_mask, torch.full_like(slot_of, -1), slot_of)
    order = torch.argsort(seat, stable=True)
    sizes = torch.bincount(seat + 1, minlength=len(ranked) + 2).tolist()          # [learner, opponents in sorted-id order, unseated]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    learner_indices = order[:sizes[0]]

    actions = torch.zeros(n, dtype=torch.long, device=dev)
    learner_log_probs = torch.zeros(0, device=dev)
    learner_values = torch.zeros(0, device=dev)
    if sizes[0]:
        learner_model.eval()                                     # stays in eval: update() switches to train() itself
        with torch.no_grad():
            out = learner_model(obs.index_select(0, learner_indices))
        picked, learner_log_probs = _draw(out.policy_logits.reshape(sizes[0], -1), legal_masks.index_select(0, learner_indices),
                                          "Learner", learner_indices)
        learner_values = (value_adapter.scalar_value_blended(out.value_logits, out.score_lead) if value_adapter is not None
                          else KataGoPPOAlgorithm.scalar_value(out.value_logits))
        actions.index_copy_(0, learner_indices, picked)

    for opp_id, model in cohort.items():                         # (the reference's order: the draws consume the generator in it)
        rank = rank_of[opp_id]
        count = sizes[rank + 1]
        if not count:
            if game_opponent is None and obs.is_cuda and n - sizes[0] > 0:
                torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)          # the seed draw of the model that does not run (see above)
            continue
        games = order[starts[rank + 1]:starts[rank + 1] + count]
        away = opponent_devices.get(opp_id) if opponent_devices is not None else \
            (lambda d: d if d != dev else None)(_device_of(model, dev))
        rows_obs, rows_mask = obs.index_select(0, games), legal_masks.index_select(0, games)
        if away is not None:                                     # league opponents may live on another card
            rows_obs, rows_mask = rows_obs.to(away), rows_mask.to(away)
        with torch.no_grad():
            out = model(rows_obs)
        picked, _ = _draw(out.policy_logits.reshape(count, -1), rows_mask, "Opponent", games)
        actions.index_copy_(0, games, picked.to(dev))

    return SplitMergeResult(actions=actions, learner_mask=learner_mask, opponent_mask=opponent_mask,
                            learner_log_probs=learner_log_probs, learner_values=learner_values, learner_indices=learner_indices)
