"""Generalised Advantage Estimation (mirror of keisei/training/gae.py:8-296).

    delta_t = r_t + gamma * nv_t * (1 - term_t) - V_t        A_t = delta_t + gamma*lam*(1 - term_t) * A_{t+1}
    nv_t    = override[t] where finite, else V_{t+1} (bootstrap `next_value` at the last step, and for the
              padded variants also at each env's own last valid step)

Four entry points with the reference's signatures.  GPU tensors run one launch of the batched
HIP scan (csrc/gae.hip, bit-identical to the reference's operation order); CPU tensors use the
vectorised host form below.  Outputs never carry autograd history; dtype follows ``values``.
"""
from __future__ import annotations

import torch

from keisei_amd import _lib


def _host_scan(rewards, values, terminated, boot, gamma, lam, override, lengths):
    """(T,N) reference-order recurrence on CPU tensors."""
    T, N = rewards.shape
    dt = values.dtype
    rewards = rewards.to(dt)
    nv = torch.empty_like(values)
    nv[:-1] = values[1:]
    nv[-1] = boot
    if lengths is not None:
        last = (lengths.to(torch.long) - 1).clamp(min=0)
        nv[last, torch.arange(N)] = boot
    if override is not None:
        ov = override.to(dt)
        nv = torch.where(torch.isnan(ov), nv, ov)
    alive = 1.0 - terminated.float()
    delta = rewards + gamma * nv * alive - values
    decay = gamma * lam * alive
    adv = torch.empty_like(values)
    run = torch.zeros(N, dtype=dt)
    for t in range(T - 1, -1, -1):
        run = delta[t] + decay[t] * run
        adv[t] = run
    return adv


def _device_scan(rewards, values, terminated, boot, gamma, lam, override, lengths):
    T, N = rewards.shape
    dt = values.dtype
    if dt not in (torch.float32, torch.float64):
        values = values.float()
        dt = torch.float32
    dev = values.device
    f64 = int(dt == torch.float64)
    r = rewards.to(device=dev, dtype=dt).contiguous()
    v = values.contiguous()
    term = terminated.to(device=dev, dtype=torch.float32).contiguous()
    nvb = boot.to(device=dev, dtype=dt).reshape(-1).contiguous()
    ov = None if override is None else override.to(device=dev, dtype=dt).contiguous()
    ln = None if lengths is None else lengths.to(device=dev, dtype=torch.long).contiguous()
    adv = torch.empty_like(v)
    _lib.call("ka_gae", r, v, term, nvb, ov, ln, adv, T, N, float(gamma), float(lam), f64, _lib.stream_ptr(dev))
    return adv


def _scan(rewards, values, terminated, boot, gamma, lam, override, lengths):
    if values.is_cuda:
        return _device_scan(rewards, values, terminated, boot, gamma, lam, override, lengths)
    return _host_scan(rewards, values, terminated, boot, gamma, lam, override, lengths)


def compute_gae(rewards: torch.Tensor, values: torch.Tensor, terminated: torch.Tensor, next_value: torch.Tensor,
                gamma: float, lam: float, next_value_override: torch.Tensor | None = None) -> torch.Tensor:
    """1-D (single trajectory, scalar ``next_value``) or 2-D ((T,N) grid, ``next_value`` (N,))."""
    with torch.no_grad():
        if rewards.ndim == 1:
            ov = None if next_value_override is None else next_value_override.unsqueeze(1)
            out = _scan(rewards.unsqueeze(1), values.unsqueeze(1), terminated.unsqueeze(1),
                        torch.as_tensor(next_value, dtype=values.dtype).reshape(1), gamma, lam, ov, None)
            return out.squeeze(1)
        return _scan(rewards, values, terminated, next_value, gamma, lam, next_value_override, None)


def compute_gae_padded(rewards: torch.Tensor, values: torch.Tensor, terminated: torch.Tensor,
                       next_values: torch.Tensor, lengths: torch.Tensor, gamma: float, lam: float,
                       next_value_override: torch.Tensor | None = None) -> torch.Tensor:
    """(T_max,N) padded trajectories; padding cells must carry terminated = 1."""
    with torch.no_grad():
        return _scan(rewards, values, terminated, next_values, gamma, lam, next_value_override, lengths)


def compute_gae_gpu(rewards: torch.Tensor, values: torch.Tensor, terminated: torch.Tensor, next_value: torch.Tensor,
                    gamma: float, lam: float, next_value_override: torch.Tensor | None = None) -> torch.Tensor:
    """(T,N) only: the flat 1-D case would chain transitions across environments."""
    if rewards.ndim != 2:
        raise ValueError(f"compute_gae_gpu only supports 2D (T, N) input, got shape {rewards.shape}")
    with torch.no_grad():
        return _scan(rewards, values, terminated, next_value, gamma, lam, next_value_override, None)


def compute_gae_padded_gpu(rewards: torch.Tensor, values: torch.Tensor, terminated: torch.Tensor,
                           next_values: torch.Tensor, lengths: torch.Tensor, gamma: float, lam: float,
                           next_value_override: torch.Tensor | None = None) -> torch.Tensor:
    if rewards.ndim != 2:
        raise ValueError(f"compute_gae_padded_gpu only supports 2D (T_max, N) input, got shape {rewards.shape}")
    with torch.no_grad():
        return _scan(rewards, values, terminated, next_values, gamma, lam, next_value_override, lengths)
