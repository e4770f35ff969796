"""KataGo-style multi-head PPO: rollout buffer, action selection and the clipped-surrogate
minibatch update (mirror of keisei/training/katago_ppo.py:19-991).

Public names, signatures, validation messages, returned metric keys and side effects follow the
reference.  Two execution paths share this front end:

* **fused HIP path** (model is an ``SEResNetModel`` on a CUDA/HIP device, optimiser is a plain
  ``torch.optim.Adam``): GAE, advantage normalisation, minibatch gather, forward, the fused
  loss+gradient kernel, the hand-written backward and the fused GradScaler/clip/Adam step all run
  as HIP kernels from libkeisei_amd.so with **no host synchronisation inside the minibatch loop**
  (the reference's NaN / zero-legal-action guards become device flags that veto the optimiser
  step and raise after the loop).
* **generic path** (CPU tensors -- the reference's whole test-suite -- or any other module /
  adapter / optimiser): the same algorithm in ordinary tensor ops.
"""
from __future__ import annotations

import dataclasses
import logging
import os
import time
from typing import Any, Callable

import torch
import torch.nn.functional as F
from torch.amp import GradScaler, autocast

from keisei_amd import _lib
from keisei_amd.training.fused_optim import FusedAdamMixin
from keisei_amd.training.gae import compute_gae_gpu
from keisei_amd.training.models.katago_base import KataGoBaseModel
from keisei_amd.training.models.se_resnet import SEResNetModel
from keisei_amd.training.value_adapter import MultiHeadValueAdapter

SCORE_NORMALIZATION = 76.0      # keisei/sl/dataset.py:32
_log = logging.getLogger(__name__)


def _amp_dtype_and_device(use_amp: bool, device: torch.device) -> tuple[torch.dtype, str]:
    """(autocast dtype, autocast device type); the dtype is a placeholder when AMP is off."""
    if use_amp and (device.type == "cpu" or torch.cuda.is_bf16_supported()):
        return torch.bfloat16, device.type
    return torch.float16, device.type


def ppo_clip_loss(new_log_probs: torch.Tensor, old_log_probs: torch.Tensor, advantages: torch.Tensor,
                  clip_epsilon: float) -> torch.Tensor:
    """-mean(min(r*A, clamp(r, 1-eps, 1+eps)*A)) with r = exp(new - old)."""
    ratio = torch.exp(new_log_probs - old_log_probs)
    clipped = torch.clamp(ratio, 1 - clip_epsilon, 1 + clip_epsilon)
    return -torch.min(ratio * advantages, clipped * advantages).mean()


def wdl_cross_entropy_loss(value_logits: torch.Tensor, value_cats: torch.Tensor) -> torch.Tensor:
    """W/D/L cross-entropy, ignore_index = -1; all-ignored batches give a graph-connected zero."""
    if not bool((value_cats >= 0).any()):
        return value_logits.sum() * 0.0
    return F.cross_entropy(value_logits, value_cats, ignore_index=-1)


def compute_value_metrics(value_logits: torch.Tensor, value_targets: torch.Tensor) -> dict[str, float]:
    pred = value_logits.argmax(dim=-1)
    frac = lambda mask: mask.float().mean().item()  # noqa: E731
    return {"value_accuracy": frac(pred == value_targets), "frac_predicted_win": frac(pred == 0),
            "frac_predicted_draw": frac(pred == 1), "frac_predicted_loss": frac(pred == 2)}


@dataclasses.dataclass(frozen=True)
class KataGoPPOParams:
    learning_rate: float = 2e-4
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_epsilon: float = 0.2
    epochs_per_batch: int = 4
    batch_size: int = 256
    lambda_policy: float = 1.0
    lambda_value: float = 1.5
    lambda_score: float = 0.02
    lambda_entropy: float = 0.01
    score_normalization: float = SCORE_NORMALIZATION
    grad_clip: float = 1.0
    use_amp: bool = False
    compile_mode: str | None = None     # accepted for config compatibility; no tracing compiler is used here
    compile_dynamic: bool = True
    entropy_decay_epochs: int = 0
    score_blend_alpha: float = 0.0
    use_terminated_for_gae: bool = True

    def __post_init__(self) -> None:
        if self.batch_size <= 0:
            raise ValueError(f"batch_size must be > 0, got {self.batch_size}")
        if self.epochs_per_batch <= 0:
            raise ValueError(f"epochs_per_batch must be > 0, got {self.epochs_per_batch}")
        if not 0.0 <= self.gamma <= 1.0:
            raise ValueError(f"gamma must be in [0, 1], got {self.gamma}")
        if not 0.0 <= self.gae_lambda <= 1.0:
            raise ValueError(f"gae_lambda must be in [0, 1], got {self.gae_lambda}")
        if self.clip_epsilon < 0.0:
            raise ValueError(f"clip_epsilon must be >= 0, got {self.clip_epsilon}")
        if self.learning_rate <= 0.0:
            raise ValueError(f"learning_rate must be > 0, got {self.learning_rate}")
        if self.grad_clip <= 0.0:
            raise ValueError(f"grad_clip must be > 0, got {self.grad_clip}")


_FIELDS = ("observations", "actions", "log_probs", "values", "rewards", "dones", "terminated", "legal_masks",
           "value_categories", "score_targets")
_COLUMN_DTYPES = {"observations": torch.float32, "actions": torch.long, "log_probs": torch.float32,
                  "values": torch.float32, "rewards": torch.float32, "dones": torch.bool, "terminated": torch.bool,
                  "legal_masks": torch.bool, "value_categories": torch.long, "score_targets": torch.float32,
                  "env_ids": torch.long, "next_value_override": torch.float32}


def _check_step_inputs(done_c, term_c, cat_c, score_c) -> None:
    """The reference's add() guards (katago_ppo.py:244-266) on host copies of one timestep's columns."""
    if bool((term_c.bool() & ~done_c.bool()).any()):
        raise AssertionError(
            "terminated must be a subset of dones: every terminated position must also be done. "
            "Got terminated=True where dones=False — likely a call site passing the merged signal.")
    bad = set(cat_c.unique().tolist()) - {-1, 0, 1, 2}
    if bad:
        raise ValueError(f"value_categories contains invalid values {bad}. "
                         f"Expected only {{-1=ignore, 0=W, 1=D, 2=L}}.")
    if bool(score_c.isnan().any()):
        raise ValueError("score_targets contains NaN. With per-step material balance, "
                         "all targets should be real-valued.")
    peak = score_c.abs().max()
    if peak > 3.5:
        raise ValueError(f"score_targets appear unnormalized: max abs value = {peak.item():.1f}. "
                         f"Expected in [-1.7, +1.7] typical, theoretical max 2.58 (guard 3.5).")


class KataGoRolloutBuffer:
    """Structure-of-arrays rollout store (katago_ppo.py:128-388): pre-allocated, doubling growth (at least
    512*num_envs rows), optional ``env_ids`` and NaN-sentinel ``next_value_override`` columns.

    Where the columns live is decided by the first ``add()`` (or the ``device`` argument): CPU tensors give the
    reference's host store; tensors on a CUDA/HIP device give the **device-resident store**  -- one
    ``ka_rollout_append`` launch per step writes the columns in HBM, packs each 11 259-byte legal mask into 352
    words and evaluates the input guards on the device, so ``add()`` reads back 16 bytes (one synchronisation
    instead of the reference's ten ``.cpu()`` calls) and ``update()`` uploads nothing.  ``KA_ROLLOUT_BUFFER=host``
    forces the host store.  Same public surface either way; ``flatten()`` of a device store returns device tensors
    (``legal_masks`` unpacked to bool on request of that call)."""

    def __init__(self, num_envs: int, obs_shape: tuple[int, ...], action_space: int,
                 device: torch.device | str | None = None) -> None:
        self.num_envs = num_envs
        self.obs_shape = obs_shape
        self.action_space = action_space
        self._alloc_samples = 0
        self._write_offset = 0
        self._step_count = 0
        self._storage: dict[str, torch.Tensor] = {}
        self._has_env_ids = False
        self._has_next_value_override = False
        self._device: torch.device | None = None if device is None else torch.device(device)
        self._flags: torch.Tensor | None = None
        self._obs_elems = 1
        for d in obs_shape:
            self._obs_elems *= int(d)

    @property
    def is_device_resident(self) -> bool:
        return self._device is not None and self._device.type == "cuda"

    def _place(self, like: torch.Tensor) -> None:
        if self._device is None:
            on_gpu = like.is_cuda and os.environ.get("KA_ROLLOUT_BUFFER", "device") != "host"
            self._device = like.device if on_gpu else torch.device("cpu")

    def _fresh(self, key: str, rows: int) -> torch.Tensor:
        dev = self._device
        if key == "observations":
            return torch.empty(rows, *self.obs_shape, device=dev)
        if key == "legal_masks":
            if self.is_device_resident:      # packed: bit j of word w = action 32 w + j
                return torch.empty(rows, (self.action_space + 31) // 32, dtype=torch.int32, device=dev)
            return torch.empty(rows, self.action_space, dtype=torch.bool)
        if key == "next_value_override":
            return torch.full((rows,), float("nan"), device=dev)
        return torch.empty(rows, dtype=_COLUMN_DTYPES[key], device=dev)

    def _ensure_capacity(self, n_samples: int) -> None:
        need = self._write_offset + n_samples
        if need <= self._alloc_samples:
            return
        cap = max(2 * need, 512 * self.num_envs)
        keys = list(_FIELDS)
        if self._has_env_ids or "env_ids" in self._storage:
            keys.append("env_ids")
        if self._has_next_value_override or "next_value_override" in self._storage:
            keys.append("next_value_override")
        grown = {k: self._fresh(k, cap) for k in keys}
        used = self._write_offset
        if used:
            for k, t in grown.items():
                if k in self._storage:
                    t[:used] = self._storage[k][:used]
        self._storage, self._alloc_samples = grown, cap

    def _column(self, key: str) -> torch.Tensor:
        if key not in self._storage:
            self._storage[key] = self._fresh(key, self._alloc_samples)
        return self._storage[key]

    @property
    def size(self) -> int:
        return self._step_count

    def clear(self) -> None:
        self._write_offset = 0
        self._step_count = 0

    def add(self, obs: torch.Tensor, actions: torch.Tensor, log_probs: torch.Tensor, values: torch.Tensor,
            rewards: torch.Tensor, dones: torch.Tensor, terminated: torch.Tensor, legal_masks: torch.Tensor,
            value_categories: torch.Tensor, score_targets: torch.Tensor, env_ids: torch.Tensor | None = None,
            next_value_override: torch.Tensor | None = None) -> None:
        """Append one timestep (n rows).  ``score_targets`` must already be normalised."""
        self._place(obs)
        if self.is_device_resident:
            self._add_device(obs, actions, log_probs, values, rewards, dones, terminated, legal_masks, value_categories,
                             score_targets, env_ids, next_value_override)
            return
        host = lambda t: t.detach().cpu()  # noqa: E731
        obs_c, act_c, lp_c, val_c, rew_c = host(obs), host(actions), host(log_probs), host(values), host(rewards)
        done_c, term_c = host(dones), host(terminated)
        mask_c, cat_c, score_c = host(legal_masks), host(value_categories), host(score_targets)
        _check_step_inputs(done_c, term_c, cat_c, score_c)
        n = obs_c.shape[0]
        if self._step_count == 0:
            self._has_env_ids = self._has_env_ids or env_ids is not None
            self._has_next_value_override = self._has_next_value_override or next_value_override is not None
        self._ensure_capacity(n)
        lo, hi = self._write_offset, self._write_offset + n
        for key, val in zip(_FIELDS, (obs_c, act_c, lp_c, val_c, rew_c, done_c, term_c, mask_c, cat_c, score_c)):
            self._storage[key][lo:hi] = val
        if env_ids is not None:
            self._column("env_ids")[lo:hi] = host(env_ids)
        if next_value_override is not None:
            self._has_next_value_override = True
            self._column("next_value_override")[lo:hi] = host(next_value_override).to(torch.float32)
        elif self._has_next_value_override and "next_value_override" in self._storage:
            self._storage["next_value_override"][lo:hi] = float("nan")     # no stale cells from a previous epoch
        self._write_offset = hi
        self._step_count += 1

    def _add_device(self, obs, actions, log_probs, values, rewards, dones, terminated, legal_masks, value_categories,
                    score_targets, env_ids, next_value_override) -> None:
        dev = self._device
        n = obs.shape[0]
        if n == 0:           # the reference's guards fail on an empty step (max() of an empty tensor): same exception here
            _check_step_inputs(dones.detach().cpu(), terminated.detach().cpu(), value_categories.detach().cpu(),
                               score_targets.detach().cpu())
        col = lambda t, dt: t.detach().to(device=dev, dtype=dt).reshape(n).contiguous()  # noqa: E731 (no-op when canonical)
        obs_c = obs.detach().to(device=dev, dtype=torch.float32).reshape(n, self._obs_elems).contiguous()
        # legal masks: bool rows (the reference's form) or the PACKED int32 rows (n, ceil(A / 32)) of the device env /
        # PendingTransitions.finalize()["legal_mask_bits"], which are copied word for word into the store's packed column
        words = (self.action_space + 31) // 32
        packed = legal_masks.dtype == torch.int32 and legal_masks.dim() == 2 and legal_masks.shape[1] == words
        if packed:
            mask_c = legal_masks.detach().to(device=dev).contiguous()
        else:
            mask_c = legal_masks.detach().to(device=dev, dtype=torch.bool).reshape(n, self.action_space).contiguous()
        act_c, cat_c = col(actions, torch.long), col(value_categories, torch.long)
        lp_c, val_c, rew_c, score_c = (col(t, torch.float32) for t in (log_probs, values, rewards, score_targets))
        done_c, term_c = col(dones, torch.bool), col(terminated, torch.bool)
        env_c = None if env_ids is None else col(env_ids, torch.long)
        ov_c = None if next_value_override is None else col(next_value_override, torch.float32)
        if self._step_count == 0:
            self._has_env_ids = self._has_env_ids or env_ids is not None
            self._has_next_value_override = self._has_next_value_override or next_value_override is not None
        self._ensure_capacity(n)
        if next_value_override is not None:
            self._has_next_value_override = True
            self._column("next_value_override")
        if env_ids is not None:
            self._column("env_ids")
        if self._flags is None:
            self._flags = torch.zeros(4, dtype=torch.int32, device=dev)
        lo, hi = self._write_offset, self._write_offset + n
        st = self._storage
        dst = lambda key: st[key][lo:hi] if key in st else None  # noqa: E731
        _lib.call("ka_rollout_append_packed" if packed else "ka_rollout_append", obs_c, mask_c, act_c, lp_c, val_c, rew_c, done_c, term_c,
                  cat_c, score_c, env_c, ov_c,
                  dst("observations"), dst("legal_masks"), dst("actions"), dst("log_probs"), dst("values"), dst("rewards"),
                  dst("dones"), dst("terminated"), dst("value_categories"), dst("score_targets"),
                  dst("env_ids") if env_ids is not None else None,
                  dst("next_value_override") if self._has_next_value_override else None,
                  self._flags, n, self._obs_elems, self.action_space, _lib.stream_ptr(dev))
        fl = self._flags.cpu()                       # the one synchronisation of add()
        if bool(fl[:3].any()) or float(fl[3:4].view(torch.float32)) > 3.5:
            self._flags.zero_()
            _check_step_inputs(done_c.cpu(), term_c.cpu(), cat_c.cpu(), score_c.cpu())     # raises the reference's message
            raise RuntimeError("rollout_append flagged a step its host check accepts")      # pragma: no cover
        self._write_offset = hi
        self._step_count += 1

    def fill_alternating_perspective_overrides(self) -> None:
        """Two-player frames alternate every ply: where no override was supplied and the transition is not
        terminal, bootstrap from -V[t+1] (katago_ppo.py:320-362).  No-op for the env_ids (split-merge) layout."""
        T, N = self._step_count, self.num_envs
        if self._has_env_ids or T <= 1 or self._write_offset != T * N:
            return
        self._has_next_value_override = True
        ov = self._column("next_value_override")[:T * N].view(T, N)
        vals = self._storage["values"][:T * N].view(T, N)
        term = self._storage["terminated"][:T * N].view(T, N).bool()
        head = ov[:-1]
        fill = torch.isnan(head) & ~term[:-1]
        head.copy_(torch.where(fill, -vals[1:], head))

    def _flatten(self, unpack_masks: bool) -> dict[str, torch.Tensor]:
        if self._step_count == 0:
            raise ValueError("Cannot flatten an empty buffer. Call add() at least once before flatten().")
        n = self._write_offset
        out = {"observations": self._storage["observations"][:n].reshape(-1, *self.obs_shape)}
        masks = self._storage["legal_masks"][:n]
        if not self.is_device_resident:
            out["legal_masks"] = masks.reshape(-1, self.action_space)
        elif unpack_masks:
            full = torch.empty(n, self.action_space, dtype=torch.bool, device=self._device)
            _lib.call("ka_unpack_mask_bits", masks, None, full, n, self.action_space, _lib.stream_ptr(self._device))
            out["legal_masks"] = full
        else:
            out["legal_bits"] = masks
        for key in _FIELDS:
            if key not in ("observations", "legal_masks"):
                out[key] = self._storage[key][:n].reshape(-1)
        if self._has_env_ids and "env_ids" in self._storage:
            out["env_ids"] = self._storage["env_ids"][:n].reshape(-1)
        if self._has_next_value_override and "next_value_override" in self._storage:
            out["next_value_override"] = self._storage["next_value_override"][:n].reshape(-1)
        return out

    def flatten(self) -> dict[str, torch.Tensor]:
        return self._flatten(unpack_masks=True)

    def flatten_packed(self) -> dict[str, torch.Tensor]:
        """``flatten()`` without materialising bool masks: a device store hands out its packed ``legal_bits`` rows
        (what ``update()`` consumes); a host store is returned as ``flatten()`` does."""
        return self._flatten(unpack_masks=False)


class _ModeBoundForward:
    """Stand-in for the reference's torch.compile wrappers (katago_ppo.py:436-459): a callable with
    ``_orig_mod`` that runs the wrapped module.  No tracing compiler is involved -- the GPU path is
    already hand-written kernels -- but callers that look for ``compiled_train`` / ``compiled_eval``
    keep working."""

    def __init__(self, module: torch.nn.Module) -> None:
        self._orig_mod = module

    def __call__(self, *args, **kwargs):
        return self._orig_mod(*args, **kwargs)


class KataGoPPOAlgorithm(FusedAdamMixin):
    def __init__(self, params: KataGoPPOParams, model: KataGoBaseModel, forward_model: torch.nn.Module | None = None,
                 warmup_epochs: int = 0, warmup_entropy: float = 0.05) -> None:
        self.params = params
        self.model = model
        self.forward_model = forward_model or model
        unwrap = lambda m: m.module if hasattr(m, "module") else m  # noqa: E731
        assert unwrap(self.forward_model) is unwrap(self.model), (
            "forward_model and model must share parameters — compile + grad clipping requires this")

        self.compiled_train: Callable[..., Any] | None = None
        self.compiled_eval: Callable[..., Any] | None = None
        if params.compile_mode is not None:
            if params.compile_mode == "reduce-overhead" and params.compile_dynamic:
                _log.warning("compile_mode='reduce-overhead' with compile_dynamic=True disables CUDA graph capture "
                             "(the main benefit of reduce-overhead). Set compile_dynamic=False for fixed-batch-size runs.")
            self.compiled_train = _ModeBoundForward(self.forward_model)
            self.compiled_eval = _ModeBoundForward(self.forward_model)
            self.forward_model.train()
            _log.info("compile_mode=%s accepted; forward passes run on hand-written HIP kernels (no tracing compiler)",
                      params.compile_mode)

        self._timing_events: dict[str, list] = {"select_actions_forward_ms": [], "update_forward_backward_ms": [],
                                                "gae_ms": []}
        self.timings: dict[str, list[float]] = {k: [] for k in self._timing_events}

        device = next(model.parameters()).device
        if hasattr(model, "configure_amp"):
            amp_dtype, amp_device = _amp_dtype_and_device(params.use_amp, device)
            model.configure_amp(enabled=params.use_amp, dtype=amp_dtype, device_type=amp_device)
            if self.compiled_train is not None:
                model._amp_frozen = True
        self.optimizer = torch.optim.Adam(model.parameters(), lr=params.learning_rate)
        self.scaler = GradScaler(enabled=params.use_amp and device.type == "cuda")
        self.warmup_epochs = warmup_epochs
        self.warmup_entropy = warmup_entropy
        self.current_entropy_coeff = params.lambda_entropy
        self._hip_state: dict[str, Any] = {}
        self.last_update_path: str | None = None       # "fused" | "generic": which step implementation update() took
        self._warned_generic: set[str] = set()

    # ------------------------------------------------------------------ small helpers
    def get_entropy_coeff(self, epoch: int) -> float:
        if epoch < self.warmup_epochs:
            return self.warmup_entropy
        span = self.params.entropy_decay_epochs
        done = epoch - self.warmup_epochs
        if span <= 0 or done >= span:
            return self.params.lambda_entropy
        return self.warmup_entropy + (done / span) * (self.params.lambda_entropy - self.warmup_entropy)

    def flush_timings(self) -> None:
        """Event pairs -> milliseconds (the only synchronisation point of the timers)."""
        for key, pairs in self._timing_events.items():
            self.timings[key] = [a.elapsed_time(b) for a, b in pairs]
            pairs.clear()

    @staticmethod
    def scalar_value(value_logits: torch.Tensor) -> torch.Tensor:
        probs = torch.softmax(value_logits, dim=-1)
        return probs[:, 0] - probs[:, 2]

    def _event_pair(self, device):
        stream = torch.cuda.current_stream(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        return a, b, stream

    # ------------------------------------------------------------------ rollout side
    @torch.no_grad()
    def select_actions(self, obs: torch.Tensor, legal_masks: torch.Tensor, value_adapter: Any | None = None,
                       ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        device = next(self.model.parameters()).device
        net = self.compiled_eval if self.compiled_eval is not None else self.forward_model
        self._set_training(False)              # = self.forward_model.eval() (katago_ppo.py:553)
        try:
            if device.type == "cuda":
                start, end, stream = self._event_pair(device)
            out = net(obs)
            if device.type == "cuda":
                end.record(stream)
                self._timing_events["select_actions_forward_ms"].append((start, end))
            raw = out.policy_logits.reshape(obs.shape[0], -1)
            if (raw.is_cuda and legal_masks.is_cuda and legal_masks.dtype == torch.bool
                    and raw.dtype in (torch.float32, torch.bfloat16) and os.environ.get("KA_SELECT_TORCH_SAMPLER", "0") != "1"):
                # the whole tail in ONE launch: masked softmax, one inverse-CDF draw per row, log-prob, the zero-legal guard and
                # the scalar value (the masked-softmax + torch.multinomial chain below is ~25 launches)
                sampled = self._sample_fused(raw, legal_masks, out, value_adapter)
                if sampled is not None:
                    return sampled
            if raw.is_cuda and legal_masks.is_cuda and legal_masks.dtype == torch.bool:
                # one launch: masked softmax + legal-action counts (the tensor-op chain below is ~12 launches)
                B_, A_ = raw.shape
                logits32 = raw.float().contiguous()
                masks_c = legal_masks.reshape(B_, A_).contiguous()
                probs = torch.empty(B_, A_, device=raw.device)
                n_legal = torch.empty(B_, dtype=torch.int32, device=raw.device)
                nan_flag = torch.zeros(1, dtype=torch.int32, device=raw.device)
                _lib.call("ka_masked_softmax", logits32, masks_c, probs, n_legal, nan_flag, B_, A_, 0, _lib.stream_ptr(raw.device))
            else:
                probs = None
                n_legal = legal_masks.sum(dim=-1)
            if bool((n_legal == 0).any()):
                empty = (n_legal == 0).nonzero(as_tuple=True)[0].tolist()
                raise RuntimeError(f"Environments {empty} have zero legal actions — "
                                   f"all-False legal mask would produce NaN")
            if probs is not None:
                actions = torch.multinomial(probs, 1, True).squeeze(1)          # what Categorical(probs).sample() draws
                log_probs = probs.gather(1, actions.unsqueeze(1)).squeeze(1).log()
            else:
                logits = raw.float().masked_fill(~legal_masks, float("-inf"))
                dist = torch.distributions.Categorical(torch.softmax(logits, dim=-1), validate_args=False)
                actions = dist.sample()
                log_probs = dist.log_prob(actions)
            if value_adapter is not None:
                values = value_adapter.scalar_value_blended(out.value_logits, out.score_lead)
            else:
                values = self.scalar_value(out.value_logits)
            return actions, log_probs, values
        finally:
            self._set_training(True)           # = self.forward_model.train()

    def _sample_fused(self, raw: torch.Tensor, legal_masks: torch.Tensor, out, value_adapter):
        """`ka_policy_sample` (loss.hip): returns (actions, log_probs, values), or None when the value adapter is not one the
        kernel knows (then the caller takes the generic route)."""
        from keisei_amd.training.value_adapter import MultiHeadValueAdapter

        if value_adapter is None:
            alpha = 0.0
        elif type(value_adapter) is MultiHeadValueAdapter:
            alpha = float(value_adapter.score_blend_alpha)
        else:
            return None
        B_, A_ = raw.shape
        dev = raw.device
        logits = raw.contiguous()
        masks_c = legal_masks.reshape(B_, A_).contiguous()
        vl = out.value_logits.float().contiguous()
        sc = out.score_lead.float().reshape(B_).contiguous() if alpha != 0.0 else None
        actions = torch.empty(B_, dtype=torch.int64, device=dev)
        log_probs = torch.empty(B_, device=dev)
        values = torch.empty(B_, device=dev)
        n_legal = torch.empty(B_, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())       # host generator: torch.manual_seed() fixes the rollout
        _lib.call("ka_policy_sample", logits, int(logits.dtype == torch.bfloat16), masks_c, 0, seed, vl, sc, alpha, actions,
                  log_probs, values, n_legal, flags, B_, A_, _lib.stream_ptr(dev))
        nan_seen, zero_legal = flags.tolist()                  # the one synchronisation of the call; both guards ride on it
        if zero_legal:                                         # katago_ppo.py:589-596
            empty = (n_legal == 0).nonzero(as_tuple=True)[0].tolist()
            raise RuntimeError(f"Environments {empty} have zero legal actions — "
                               f"all-False legal mask would produce NaN")
        if nan_seen:
            # the reference's softmax + Categorical(probs).sample() refuses NaN probabilities (torch.multinomial raises);
            # a diverged model must not keep stepping the environments and filling the buffer
            raise RuntimeError("NaN in raw policy logits in select_actions — probability tensor contains nan "
                               "(the model has diverged)")
        return actions, log_probs, values

    def _set_training(self, mode: bool) -> None:
        """forward_model.train(mode) without walking the module tree on every rollout step: nn.Module.train() recurses
        through ~450 modules with nn.Module.__setattr__ at each (about a millisecond per select_actions call at 40
        blocks, twice per call).  The flat module list is cached for as long as no module anywhere registered a
        submodule / parameter / buffer (`_structure.structure_version`: convert_sync_batchnorm, a replaced head, a rebuilt
        model all bump it); a tree in which any module overrides train() -- or carries hooks on it -- takes the plain
        `fm.train(mode)` (DistributedDataParallel's override only forwards to nn.Module.train and is exempt)."""
        from torch.nn.parallel import DistributedDataParallel
        from keisei_amd.training._structure import structure_version

        fm = self.forward_model
        ver = structure_version()
        ent = getattr(self, "_mode_cache", None)
        if ent is None or ent[0] is not fm or ent[2] != ver:
            mods = list(fm.modules())
            plain = all(type(m).train is torch.nn.Module.train or type(m) is DistributedDataParallel for m in mods)
            ent = self._mode_cache = (fm, mods if plain else None, ver)
        if ent[1] is None:
            fm.train(mode)
            return
        for mod in ent[1]:
            mod.__dict__["training"] = mode

    # ------------------------------------------------------------------ advantages
    def _advantages(self, data, buffer, next_values, device) -> torch.Tensor:
        """GAE over the buffer layout (grid / per-env / flat) (katago_ppo.py:651-773).  The result lives where the
        rollout columns live: on the device for a device-resident store, on the CPU otherwise."""
        from keisei_amd.training.gae import compute_gae  # resolved at call time (tests patch the module attribute)

        p = self.params
        T, N = buffer.size, buffer.num_envs
        total = data["rewards"].numel()
        key = "terminated" if p.use_terminated_for_gae else "dones"
        resident = data["rewards"].is_cuda
        if total == T * N:
            grid = lambda name: data[name].reshape(T, N)  # noqa: E731
            rewards, values, term = grid("rewards").float(), grid("values").float(), grid(key)
            ov = grid("next_value_override").float() if "next_value_override" in data else None
            if device.type == "cuda":
                start, end, stream = self._event_pair(device)
                adv = compute_gae_gpu(rewards.to(device), values.to(device), term.to(device),
                                      next_values.detach().float().to(device), gamma=p.gamma, lam=p.gae_lambda,
                                      next_value_override=None if ov is None else ov.to(device)).reshape(-1)
                end.record(stream)
                self._timing_events["gae_ms"].append((start, end))
                return adv if resident else adv.cpu()
            return compute_gae(rewards, values, term, next_values.detach().float().cpu(), gamma=p.gamma, lam=p.gae_lambda,
                               next_value_override=ov).reshape(-1)
        if "env_ids" in data and resident:
            return self._advantages_per_env_device(data, next_values, key, device)
        boot_cpu = next_values.detach().float().cpu()
        if "env_ids" in data:
            from keisei_amd.training.gae import compute_gae_padded

            env_ids = data["env_ids"]
            order = torch.argsort(env_ids, stable=True)
            envs, counts = env_ids[order].unique_consecutive(return_counts=True)
            lengths = counts.tolist()
            groups = torch.split(order, lengths)
            n_env, t_max = len(lengths), max(lengths)
            if envs.max() >= boot_cpu.shape[0]:
                raise IndexError(f"env_id {envs.max().item()} >= next_values size {boot_cpu.shape[0]}")
            r_pad, v_pad = torch.zeros(t_max, n_env), torch.zeros(t_max, n_env)
            term_pad = torch.ones(t_max, n_env)                   # padding = terminated, zeroes propagation
            ov_pad = torch.full((t_max, n_env), float("nan")) if "next_value_override" in data else None
            boot = torch.zeros(n_env)
            for j, rows in enumerate(groups):
                L = lengths[j]
                r_pad[:L, j], v_pad[:L, j] = data["rewards"][rows], data["values"][rows]
                term_pad[:L, j] = data[key][rows]
                boot[j] = boot_cpu[envs[j]]
                if ov_pad is not None:
                    ov_pad[:L, j] = data["next_value_override"][rows].float()
            len_t = torch.tensor(lengths)
            if device.type == "cuda":
                from keisei_amd.training.gae import compute_gae_padded_gpu

                padded = compute_gae_padded_gpu(r_pad.to(device), v_pad.to(device), term_pad.to(device), boot.to(device),
                                                len_t, gamma=p.gamma, lam=p.gae_lambda,
                                                next_value_override=None if ov_pad is None else ov_pad.to(device)).cpu()
            else:
                padded = compute_gae_padded(r_pad, v_pad, term_pad, boot, len_t, gamma=p.gamma, lam=p.gae_lambda,
                                            next_value_override=ov_pad)
            adv = torch.zeros(total)
            for j, rows in enumerate(groups):
                adv[rows] = padded[:lengths[j], j]
            return adv
        # legacy flat layout: one chain, bootstrap from the mean next value
        return compute_gae(data["rewards"].float(), data["values"].float(), data[key], boot_cpu.mean(),
                           gamma=p.gamma, lam=p.gae_lambda)

    def _advantages_per_env_device(self, data, next_values, key, device) -> torch.Tensor:
        """The per-environment (split-merge) layout for a device-resident store: the same padded (T_max, n_env) grids
        as above, built with index arithmetic on the device instead of a Python loop over environments; the scan is
        the same ``ka_gae`` launch.  One host read (T_max, n_env, largest env id)."""
        from keisei_amd.training.gae import compute_gae_padded_gpu

        p = self.params
        env_ids = data["env_ids"]
        total = env_ids.numel()
        order = torch.argsort(env_ids, stable=True)
        envs, counts = env_ids[order].unique_consecutive(return_counts=True)
        t_max, n_env, top = (int(v) for v in torch.stack([counts.max(), counts.new_tensor(counts.numel()), envs.max()]).tolist())
        boot_all = next_values.detach().float().to(device).reshape(-1)
        if top >= boot_all.shape[0]:
            raise IndexError(f"env_id {top} >= next_values size {boot_all.shape[0]}")
        starts = torch.cumsum(counts, 0) - counts
        col = torch.repeat_interleave(torch.arange(n_env, device=device), counts, output_size=total)
        pos = torch.arange(total, device=device) - starts[col]          # step index inside its environment
        cell = pos * n_env + col                                        # (t, env) cell of the padded grids
        scatter = lambda fill, src: torch.full((t_max * n_env,), fill, device=device).index_put_(  # noqa: E731
            (cell,), src[order].float()).view(t_max, n_env)
        r_pad, v_pad = scatter(0.0, data["rewards"]), scatter(0.0, data["values"])
        term_pad = scatter(1.0, data[key])                             # padding = terminated, zeroes propagation
        ov_pad = scatter(float("nan"), data["next_value_override"]) if "next_value_override" in data else None
        padded = compute_gae_padded_gpu(r_pad, v_pad, term_pad, boot_all[envs], counts, gamma=p.gamma, lam=p.gae_lambda,
                                        next_value_override=ov_pad)
        adv = torch.empty(total, device=device)
        adv[order] = padded.reshape(-1)[cell]
        return adv

    # ------------------------------------------------------------------ update
    def update(self, buffer: KataGoRolloutBuffer, next_values: torch.Tensor, value_adapter: Any | None = None,
               heartbeat_fn: Any | None = None) -> dict[str, float]:
        self.forward_model.train()
        assert self.forward_model.training, (
            "forward_model must be in train mode at start of update() — compiled_train graph requires this")
        self._timing_events["update_forward_backward_ms"].clear()
        self._timing_events["gae_ms"].clear()

        packed = getattr(buffer, "flatten_packed", None)        # device-resident store: packed mask rows, no host copies
        data = packed() if packed is not None else buffer.flatten()
        self._n_actions = getattr(buffer, "action_space", None)
        total = data["rewards"].numel()
        device = next(self.model.parameters()).device
        advantages = self._advantages(data, buffer, next_values, device)
        if advantages.numel() > 1:
            if advantages.is_cuda:
                src = advantages.float().contiguous()
                advantages = torch.empty_like(src)
                _lib.call("ka_normalize_advantages", src, advantages, src.numel(), _lib.stream_ptr(src.device))
            else:
                advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
        batch = min(self.params.batch_size, total)

        blocker = self._fused_path_blocker(device, value_adapter)
        if blocker is None:
            self.last_update_path = "fused"
            with torch.cuda.device(device):
                metrics = self._update_fused(data, advantages, total, batch, device, value_adapter, heartbeat_fn)
        else:
            if device.type == "cuda":
                if os.environ.get("KEISEI_AMD_STRICT", "0") == "1":
                    raise _lib.KeiseiHipError(f"update() cannot take the fused HIP step: {blocker} (KEISEI_AMD_STRICT=1)")
                if blocker not in self._warned_generic:
                    self._warned_generic.add(blocker)
                    _log.warning("update() on %s runs the generic torch-op step, not the fused HIP step: %s", device, blocker)
            self.last_update_path = "generic"
            metrics = self._update_generic(data, advantages, total, batch, device, value_adapter, heartbeat_fn)
        buffer.clear()
        self.forward_model.train()
        return metrics

    def _action_space(self, data) -> int:
        """Width of an unpacked legal-mask row of the epoch being updated."""
        if "legal_masks" in data:
            return int(data["legal_masks"].shape[1])
        if self._n_actions is None:
            raise ValueError("a buffer that hands out packed legal_bits must expose action_space")
        return int(self._n_actions)

    # ---- generic path -------------------------------------------------------------------
    def _update_generic(self, data, advantages, total, batch, device, value_adapter, heartbeat_fn):
        p = self.params
        amp_dtype, amp_device = _amp_dtype_and_device(p.use_amp, device)
        side = None
        if "legal_bits" in data:                                   # device-resident store: unpack once for the tensor ops
            obs = data["observations"].to(device)
            masks = torch.empty(total, self._action_space(data), dtype=torch.bool, device=data["legal_bits"].device)
            _lib.call("ka_unpack_mask_bits", data["legal_bits"], None, masks, total, masks.shape[1],
                      _lib.stream_ptr(masks.device))
            masks = masks.to(device)
        elif device.type == "cuda" and not data["observations"].is_cuda:
            side = torch.cuda.Stream(device)
            with torch.cuda.stream(side):
                obs = data["observations"].pin_memory().to(device, non_blocking=True)
                masks = data["legal_masks"].pin_memory().to(device, non_blocking=True)
        else:
            obs, masks = data["observations"].to(device), data["legal_masks"].to(device)
        move = lambda t: t.to(device, non_blocking=True)  # noqa: E731
        actions, old_lp, adv = move(data["actions"]), move(data["log_probs"]), move(advantages)
        cats, score_t = move(data["value_categories"]), move(data["score_targets"])
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)

        acc = {k: torch.zeros((), device=device) for k in ("policy", "value", "score", "entropy", "gnorm")}
        zero = torch.zeros((), device=device)
        n_updates = 0
        last_logits = last_cats = None
        net = self.compiled_train if self.compiled_train is not None else self.forward_model
        for _ in range(p.epochs_per_batch):
            perm = torch.randperm(total, device=device)
            for lo in range(0, total, batch):
                idx = perm[lo:lo + batch]
                b_obs, b_mask, b_cats = obs[idx], masks[idx], cats[idx]
                if device.type == "cuda":
                    start, end, stream = self._event_pair(device)
                with autocast(device_type=amp_device, dtype=amp_dtype, enabled=p.use_amp):
                    out = net(b_obs)
                    logits = out.policy_logits.reshape(b_obs.shape[0], -1)
                    if bool(logits.isnan().any()):
                        raise RuntimeError("NaN in raw policy logits from model forward pass")
                    if bool((b_mask.sum(dim=-1) == 0).any()):
                        raise RuntimeError("Batch contains samples with zero legal actions in update(). "
                                           "Check that terminal-state masks are not stored in the buffer.")
                    logp = F.log_softmax(logits.masked_fill(~b_mask, float("-inf")), dim=-1)
                    new_lp = logp.gather(1, actions[idx].unsqueeze(1)).squeeze(1)
                    policy_loss = ppo_clip_loss(new_lp, old_lp[idx], adv[idx], p.clip_epsilon)
                    entropy = -(logp.exp() * logp.masked_fill(~b_mask, 0.0)).sum(dim=-1).mean()
                    if value_adapter is not None:
                        vs_loss = value_adapter.compute_value_loss(out.value_logits, returns=None, value_cats=b_cats,
                                                                   score_targets=score_t[idx], score_pred=out.score_lead)
                        value_loss, score_loss = vs_loss, zero
                    else:
                        value_loss = wdl_cross_entropy_loss(out.value_logits, b_cats)
                        score_loss = F.mse_loss(out.score_lead.squeeze(-1), score_t[idx])
                        vs_loss = p.lambda_value * value_loss + p.lambda_score * score_loss
                    loss = p.lambda_policy * policy_loss + vs_loss - self.current_entropy_coeff * entropy
                self.optimizer.zero_grad(set_to_none=True)
                self.scaler.scale(loss).backward()
                self.scaler.unscale_(self.optimizer)
                gnorm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), p.grad_clip)
                self.scaler.step(self.optimizer)
                self.scaler.update()
                self._notify_weights_changed()
                if device.type == "cuda":
                    end.record(stream)
                    self._timing_events["update_forward_backward_ms"].append((start, end))
                acc["policy"] += policy_loss.detach(); acc["value"] += value_loss.detach()
                acc["score"] += score_loss.detach(); acc["entropy"] += entropy.detach()
                acc["gnorm"] += gnorm.detach() if isinstance(gnorm, torch.Tensor) else float(gnorm)
                n_updates += 1
                last_logits, last_cats = out.value_logits.detach(), b_cats
                if heartbeat_fn is not None:
                    heartbeat_fn()
        d = max(n_updates, 1)
        metrics = {"policy_loss": (acc["policy"] / d).item(), "value_loss": (acc["value"] / d).item(),
                   "score_loss": (acc["score"] / d).item(), "entropy": (acc["entropy"] / d).item(),
                   "gradient_norm": (acc["gnorm"] / d).item()}
        if last_logits is not None:
            valid = last_cats >= 0
            if bool(valid.any()):
                metrics.update(compute_value_metrics(last_logits[valid], last_cats[valid]))
        return metrics

    def _notify_weights_changed(self) -> None:
        eng = getattr(self.model, "_hip_engine", None)
        if eng is not None:
            eng.notify_weights_updated()

    # ---- fused HIP path -----------------------------------------------------------------
    def _fused_path_blocker(self, device, value_adapter) -> str | None:
        """None when ``update()`` can run as the fused HIP step; otherwise the reason it cannot.  The fused step covers
        what the reference's training loop uses (katago_loop.py:552-559: an SEResNetModel, a plain single-group Adam,
        and either no adapter or a MultiHeadValueAdapter).  Anything else needs user Python inside the step (a custom
        adapter's ``compute_value_loss``, another optimiser's ``step``) and runs the reference's statement sequence on
        torch ops -- explicitly: logged once per reason, visible as ``last_update_path``, and refused when
        ``KEISEI_AMD_STRICT=1``."""
        if device.type != "cuda":
            return "model is not on a GPU"
        if not isinstance(self.model, SEResNetModel):
            return f"model is {type(self.model).__name__}, not SEResNetModel"
        if value_adapter is not None and type(value_adapter) is not MultiHeadValueAdapter:
            return f"value adapter {type(value_adapter).__name__} is not MultiHeadValueAdapter (its loss is user Python)"
        if not self._fused_optimizer_ok():
            return "optimizer is not a plain single-group fp32 torch.optim.Adam (weight decay / amsgrad / param groups)"
        return None

    def _fused_path_available(self, device, value_adapter) -> bool:
        return self._fused_path_blocker(device, value_adapter) is None

    def _fused_begin(self, dataset: dict, device, value_adapter) -> dict:
        """Device-side state of one fused update: epoch dataset tensors (already on `device`), loss weights,
        Adam tables, GradScaler mirror, metric accumulators."""
        p = self.params
        st = self._adam_tables(device)
        scaler_t = None
        if self.scaler.is_enabled():
            if self.scaler._scale is None:
                self.scaler._lazy_init_scale_growth_tracker(device)
            scaler_t = torch.stack([self.scaler._scale.float().reshape(()),
                                    self.scaler._growth_tracker.float().reshape(())])
        ddp = reducer = None
        fm = self.forward_model
        if (isinstance(fm, torch.nn.parallel.DistributedDataParallel) and torch.distributed.is_initialized()
                and (torch.distributed.get_world_size(fm.process_group) > 1 or os.environ.get("KA_FORCE_COLLECTIVES", "0") == "1")
                and os.environ.get("KA_DDP_OVERLAP", "1") != "0"):
            from keisei_amd.hip.grad_reducer import OverlappedGradReducer
            ddp, reducer = fm, OverlappedGradReducer(fm.process_group)
        return {
            "ddp": ddp, "reducer": reducer,
            "data": dataset, "st": st, "scaler_t": scaler_t,
            "gscale": scaler_t[0:1] if scaler_t is not None else None,
            "lam_v": value_adapter.lambda_value if value_adapter is not None else p.lambda_value,
            "lam_s": value_adapter.lambda_score if value_adapter is not None else p.lambda_score,
            "combined": int(value_adapter is not None),
            "acc": torch.zeros(5, device=device),          # policy, value, score, entropy, grad-norm sums
            "flags": torch.zeros(2, dtype=torch.int32, device=device),
            "out_m": torch.zeros(16, device=device), "n_updates": 0,
        }

    def _fused_step(self, fs: dict, idx: torch.Tensor, device) -> None:
        """One PPO minibatch: fused gather+forward, loss+gradient kernel, hand-written backward, clip+Adam.
        Pure launches -- nothing here waits for the GPU."""
        p = self.params
        call, sp = _lib.call, _lib.stream_ptr(device)
        d, st = fs["data"], fs["st"]
        group = self.optimizer.param_groups[0]
        beta1, beta2 = group["betas"]
        B, A = idx.shape[0], d["n_actions"]
        ht = fs.get("host_ms")                  # optional host-side (enqueue) timing per phase, no GPU sync (bench diagnostics)
        t0 = time.perf_counter() if ht is not None else 0.0
        ddp = fs.get("ddp")
        if ddp is not None:
            # DistributedDataParallel decides in ITS FORWARD whether its reducer will run in the coming backward
            # (prepare_for_backward): the forward must already be under no_sync(), or DDP copies every gradient into its own
            # buckets (one scaled-copy launch per parameter) and all-reduces them a second time beside the engine's exchange
            with ddp.no_sync():
                out = self.forward_model(d["obs"], gather_idx=idx)
        else:
            out = self.forward_model(d["obs"], gather_idx=idx)
        t1 = time.perf_counter() if ht is not None else 0.0
        logits = out.policy_logits.reshape(B, A)
        dlogits = torch.empty_like(logits)
        new_lp = torch.empty(B, device=device); rowloss = torch.empty(B, device=device); rowent = torch.empty(B, device=device)
        dv = torch.empty(B, 3, device=device); ds = torch.empty(B, 1, device=device)
        call("ka_policy_loss", logits, d["masks"], d["actions"], d["old_lp"], d["adv"], idx, dlogits, new_lp, rowloss,
             rowent, fs["flags"], fs["gscale"], float(p.clip_epsilon), float(p.lambda_policy) / B,
             float(self.current_entropy_coeff) / B, B, A, d["mask_words"], sp)
        call("ka_value_loss", out.value_logits, out.score_lead, d["cats"], d["score_t"], idx, rowloss, rowent, dv, ds,
             fs["out_m"], fs["acc"], fs["gscale"], float(p.lambda_policy), float(fs["lam_v"]), float(fs["lam_s"]),
             float(self.current_entropy_coeff), fs["combined"], B, sp)
        self.optimizer.zero_grad(set_to_none=True)
        t2 = time.perf_counter() if ht is not None else 0.0
        if ddp is not None:
            # the engine exchanges gradient buckets DURING its backward (hip/grad_reducer.py); DDP's own post-backward
            # reduction is switched off for this pass
            engine = self.model._hip_engine
            engine.grad_reducer = fs["reducer"]
            try:
                with ddp.no_sync():
                    torch.autograd.backward([out.policy_logits, out.value_logits, out.score_lead],
                                            [dlogits.view_as(out.policy_logits), dv, ds])
            finally:
                engine.grad_reducer = None
        else:
            torch.autograd.backward([out.policy_logits, out.value_logits, out.score_lead],
                                    [dlogits.view_as(out.policy_logits), dv, ds])
        t3 = time.perf_counter() if ht is not None else 0.0
        tab = self._upload_table(st, device)
        call("ka_clip_adam_step", tab, st["blk_t"], st["blk_o"], st["nblocks"], st["partial"], st["ctl"],
             st["step_dev"], fs["scaler_t"], fs["flags"], fs["acc"][4:5], float(p.grad_clip), float(group["lr"]),
             float(beta1), float(beta2), float(group["eps"]), sp)
        if ht is not None:
            t4 = time.perf_counter()
            for k, v in (("forward", t1 - t0), ("loss", t2 - t1), ("backward", t3 - t2), ("optimizer", t4 - t3)):
                ht[k] = ht.get(k, 0.0) + 1e3 * v
        self._notify_weights_changed()
        fs["n_updates"] += 1

    def _fused_end(self, fs: dict) -> dict[str, float]:
        """The single host synchronisation of an update: metrics, optimiser step counters, scaler, guards."""
        st, scaler_t = fs["st"], fs["scaler_t"]
        host = torch.cat([fs["acc"], fs["out_m"][:9], st["step_dev"], fs["flags"].float()]).cpu().tolist()
        sums, last, step_now, fl = host[:5], host[5:14], host[14], host[15:17]
        for q in st["params"]:
            self.optimizer.state[q]["step"].fill_(step_now)
        if scaler_t is not None:
            self.scaler._scale.copy_(scaler_t[0]); self.scaler._growth_tracker.copy_(scaler_t[1].to(torch.int32))
        if fl[0]:
            raise RuntimeError("NaN in raw policy logits from model forward pass")
        if int(fl[1]) & 2:
            raise RuntimeError("update(): an action index lies outside [0, action_space) (index out of bounds in gather)")
        if int(fl[1]) & 1:
            raise RuntimeError("Batch contains samples with zero legal actions in update(). "
                               "Check that terminal-state masks are not stored in the buffer.")
        d = max(fs["n_updates"], 1)
        metrics = {"policy_loss": sums[0] / d, "value_loss": sums[1] / d, "score_loss": sums[2] / d,
                   "entropy": sums[3] / d, "gradient_norm": sums[4] / d}
        if last[5] > 0:      # valid W/D/L labels in the last minibatch
            metrics.update({"value_accuracy": last[6], "frac_predicted_win": last[7], "frac_predicted_draw": last[8],
                            "frac_predicted_loss": max(0.0, 1.0 - last[7] - last[8])})
        return metrics

    def _update_fused(self, data, advantages, total, batch, device, value_adapter, heartbeat_fn):
        p = self.params
        move = lambda t: t.to(device, non_blocking=True)  # noqa: E731
        if "legal_bits" in data:                      # device-resident store: the epoch is already in HBM, masks packed
            obs, masks = move(data["observations"]), move(data["legal_bits"])
            words, n_actions = masks.shape[1], self._action_space(data)
        else:
            side = torch.cuda.Stream(device)
            with torch.cuda.stream(side):
                obs = data["observations"].pin_memory().to(device, non_blocking=True)
                masks = data["legal_masks"].pin_memory().to(device, non_blocking=True)
            torch.cuda.current_stream(device).wait_stream(side)
            words, n_actions = 0, masks.shape[1]
        # the kernels read raw pointers: fix every column's dtype and layout here (a caller's int32 actions or
        # non-bool masks would otherwise be misread, not rejected)
        col = lambda t, dt: move(t.to(dt)).contiguous()  # noqa: E731
        if words == 0 and masks.dtype != torch.bool:
            masks = masks.to(torch.bool)
        dataset = {"obs": obs.float().contiguous(), "masks": masks.contiguous(), "mask_words": words, "n_actions": n_actions,
                   "actions": col(data["actions"], torch.int64), "old_lp": col(data["log_probs"], torch.float32),
                   "adv": col(advantages, torch.float32), "cats": col(data["value_categories"], torch.int64),
                   "score_t": col(data["score_targets"], torch.float32)}
        fs = self._fused_begin(dataset, device, value_adapter)
        for _ in range(p.epochs_per_batch):
            perm = torch.randperm(total, device=device)
            for lo in range(0, total, batch):
                start, end, stream = self._event_pair(device)
                self._fused_step(fs, perm[lo:lo + batch], device)
                end.record(stream)
                self._timing_events["update_forward_backward_ms"].append((start, end))
                if heartbeat_fn is not None:
                    heartbeat_fn()
        return self._fused_end(fs)
