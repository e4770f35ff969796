"""Supervised-learning trainer for the KataGo-contract models (API mirror of keisei/sl/trainer.py:19-191).

``SLConfig`` / ``SLTrainer(model, config).train_epoch() -> {"policy_loss", "value_loss", "score_loss"}`` keep the
reference's names, defaults, validation and side effects (Adam, GradScaler, one CosineAnnealingLR tick per epoch that
trained on data).  Two execution paths:

* **fused HIP path** (``SEResNetModel`` on a CUDA/HIP device, fp32 or bf16 AMP): the same forward / backward kernels
  as the PPO update, ``ka_policy_ce`` (cross-entropy over the 11 259 actions + its gradient in one pass over the
  logits) and ``ka_value_loss`` (W/D/L cross-entropy, score MSE, the batch means) instead of the reference's three
  loss ops and their autograd, and the fused GradScaler/clip/Adam launch.  Batches are gathered from the shard maps
  by ``SLDataset.read_batch`` on a helper thread one batch ahead and uploaded from pinned memory; nothing in the
  loop waits for the GPU -- the epoch's sums are read back once at the end.
* **generic path** (CPU tensors, other models): the reference's loop in ordinary tensor ops.

The shuffling is the reference's: the batch order comes from a ``DataLoader`` (``shuffle=True``) -- over the items on
the generic path, over their indices on the fused path -- so a seeded run visits the positions in the same order.
"""
from __future__ import annotations

import logging
import math
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from pathlib import Path

import torch
import torch.nn.functional as F
from torch.amp import GradScaler, autocast
from torch.utils.data import DataLoader, Dataset, get_worker_info

from keisei_amd import _lib
from keisei_amd.sl.dataset import SLDataset
from keisei_amd.training.fused_optim import FusedAdamMixin
from keisei_amd.training.models.katago_base import KataGoBaseModel
from keisei_amd.training.models.se_resnet import SEResNetModel

logger = logging.getLogger(__name__)


@dataclass
class SLConfig:
    data_dir: str
    batch_size: int = 4096
    learning_rate: float = 1e-3
    total_epochs: int = 30
    num_workers: int = 0
    lambda_policy: float = 1.0
    lambda_value: float = 1.5
    lambda_score: float = 0.02
    grad_clip: float = 0.5
    use_amp: bool = False
    allow_placeholder: bool = False

    def __post_init__(self) -> None:
        checks = (("grad_clip", self.grad_clip > 0, "> 0"), ("total_epochs", self.total_epochs >= 0, ">= 0"),
                  ("batch_size", self.batch_size > 0, "> 0"), ("learning_rate", self.learning_rate > 0, "> 0"),
                  ("num_workers", self.num_workers >= 0, ">= 0"))
        for name, ok, bound in checks:
            if not ok:
                raise ValueError(f"{name} must be {bound}, got {getattr(self, name)}")
        # a zero weight switches a head off; a negative one would ascend on it, NaN/inf poison the sum (trainer.py:44-54)
        for name in ("lambda_policy", "lambda_value", "lambda_score"):
            value = getattr(self, name)
            if not math.isfinite(value):
                raise ValueError(f"{name} must be finite, got {value!r}")
            if value < 0:
                raise ValueError(f"{name} must be >= 0, got {value!r}")


def _sl_worker_init(worker_id: int) -> None:
    """DataLoader workers reopen the shard maps instead of sharing the parent's (trainer.py:60-71)."""
    info = get_worker_info()
    if info is None:
        return
    ds = info.dataset
    while hasattr(ds, "dataset"):
        ds = ds.dataset
    if hasattr(ds, "clear_cache"):
        ds.clear_cache()


class _Indices(Dataset):
    """0 .. n-1: the sampler machinery of a DataLoader without the per-item decoding."""

    def __init__(self, n: int) -> None:
        self.n = n

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int) -> int:
        return i


class SLTrainer(FusedAdamMixin):
    """Trains one epoch per ``train_epoch()`` call; checkpointing is the caller's business."""

    def __init__(self, model: KataGoBaseModel, config: SLConfig) -> None:
        self.model = model
        self.config = config
        self.device = next(model.parameters()).device
        self.optimizer = torch.optim.Adam(model.parameters(), lr=config.learning_rate)
        on_gpu = self.device.type == "cuda"
        self.scaler = GradScaler(enabled=config.use_amp and on_gpu)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=max(config.total_epochs, 1),
                                                                    eta_min=1e-6)
        if config.use_amp and (self.device.type == "cpu" or torch.cuda.is_bf16_supported()):
            self._amp_dtype = torch.bfloat16
        else:
            self._amp_dtype = torch.float16          # unused placeholder when AMP is off
        self._amp_device_type = self.device.type
        model.configure_amp(enabled=config.use_amp, dtype=self._amp_dtype, device_type=self._amp_device_type)
        self.dataset = SLDataset(Path(config.data_dir), allow_placeholder=config.allow_placeholder)
        has_data = len(self.dataset) > 0
        workers = config.num_workers if has_data else 0
        self.dataloader = DataLoader(self.dataset, batch_size=config.batch_size, shuffle=has_data, num_workers=workers,
                                     pin_memory=on_gpu and workers > 0, persistent_workers=workers > 0,
                                     worker_init_fn=_sl_worker_init if workers > 0 else None)
        self._index_loader = DataLoader(_Indices(len(self.dataset)), batch_size=config.batch_size, shuffle=has_data)
        self._hip_state: dict = {}

    # ------------------------------------------------------------------ dispatch
    def _fused_path_available(self) -> bool:
        if self.device.type != "cuda" or not isinstance(self.model, SEResNetModel):
            return False
        if self.config.use_amp and self._amp_dtype != torch.bfloat16:
            return False
        return self._fused_optimizer_ok()

    def train_epoch(self) -> dict[str, float]:
        self.model.train()
        if self._fused_path_available():
            sums, batches = self._epoch_fused()
        else:
            sums, batches = self._epoch_generic()
        if batches > 0:                       # an empty dataset must not burn annealing ticks (trainer.py:176-179)
            self.scheduler.step()
        d = max(batches, 1)
        metrics = {"policy_loss": sums[0] / d, "value_loss": sums[1] / d, "score_loss": sums[2] / d}
        logger.info("SL epoch | policy=%.4f value=%.4f score=%.4f lr=%.6f", metrics["policy_loss"], metrics["value_loss"],
                    metrics["score_loss"], self.optimizer.param_groups[0]["lr"])
        return metrics

    # ------------------------------------------------------------------ generic path (trainer.py:133-174)
    def _epoch_generic(self):
        cfg = self.config
        sums = [0.0, 0.0, 0.0]
        batches = 0
        for batch in self.dataloader:
            obs = batch["observation"].to(self.device)
            tp, tv, ts = (batch[k].to(self.device) for k in ("policy_target", "value_target", "score_target"))
            out = self.model(obs)
            with autocast(device_type=self._amp_device_type, dtype=self._amp_dtype, enabled=cfg.use_amp):
                policy_loss = F.cross_entropy(out.policy_logits.reshape(obs.shape[0], -1), tp)
                value_loss = F.cross_entropy(out.value_logits, tv)
                score_loss = F.mse_loss(out.score_lead.squeeze(-1), ts)
                loss = cfg.lambda_policy * policy_loss + cfg.lambda_value * value_loss + cfg.lambda_score * score_loss
            self.optimizer.zero_grad(set_to_none=True)
            self.scaler.scale(loss).backward()
            self.scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), cfg.grad_clip)
            self.scaler.step(self.optimizer)
            self.scaler.update()
            eng = getattr(self.model, "_hip_engine", None)
            if eng is not None:
                eng.notify_weights_updated()
            for i, v in enumerate((policy_loss, value_loss, score_loss)):
                sums[i] += v.item()
            batches += 1
        return sums, batches

    # ------------------------------------------------------------------ fused HIP path
    def _epoch_fused(self):
        cfg, dev = self.config, self.device
        call, sp = _lib.call, _lib.stream_ptr(dev)
        st = self._adam_tables(dev)
        scaler_t = None
        if self.scaler.is_enabled():
            if self.scaler._scale is None:
                self.scaler._lazy_init_scale_growth_tracker(dev)
            scaler_t = torch.stack([self.scaler._scale.float().reshape(()), self.scaler._growth_tracker.float().reshape(())])
        gscale = scaler_t[0:1] if scaler_t is not None else None
        acc = torch.zeros(5, device=dev)               # sums of policy / value / score / (entropy, unused) / grad norm
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        out_m = torch.zeros(16, device=dev)
        group = self.optimizer.param_groups[0]
        beta1, beta2 = group["betas"]
        A = None
        batches = 0
        copy_stream = torch.cuda.Stream(dev)
        main = torch.cuda.current_stream(dev)

        def fetch(indices):
            host = self.dataset.read_batch(indices.tolist(), pin=True)
            with torch.cuda.stream(copy_stream):
                devb = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            return host, devb, ready

        order = iter(self._index_loader)
        with ThreadPoolExecutor(max_workers=1) as pool:
            first = next(order, None)
            pending = pool.submit(fetch, first) if first is not None else None
            while pending is not None:
                host, batch, ready = pending.result()
                nxt = next(order, None)
                pending = pool.submit(fetch, nxt) if nxt is not None else None
                main.wait_event(ready)
                for t in batch.values():
                    t.record_stream(main)
                obs = batch["observation"]
                B = obs.shape[0]
                out = self.model(obs)
                logits = out.policy_logits.reshape(B, -1)
                A = logits.shape[1]
                dlogits = torch.empty_like(logits)
                rowloss = torch.empty(B, device=dev)
                rowent = torch.zeros(B, device=dev)
                dv, ds = torch.empty(B, 3, device=dev), torch.empty(B, 1, device=dev)
                call("ka_policy_ce", logits, batch["policy_target"], None, dlogits, rowloss, flags, gscale,
                     float(cfg.lambda_policy) / B, B, A, sp)
                call("ka_value_loss", out.value_logits, out.score_lead, batch["value_target"], batch["score_target"], None,
                     rowloss, rowent, dv, ds, out_m, acc, gscale, float(cfg.lambda_policy), float(cfg.lambda_value),
                     float(cfg.lambda_score), 0.0, 0, B, sp)
                self.optimizer.zero_grad(set_to_none=True)
                torch.autograd.backward([out.policy_logits, out.value_logits, out.score_lead],
                                        [dlogits.view_as(out.policy_logits), dv, ds])
                tab = self._upload_table(st, dev)
                call("ka_clip_adam_step", tab, st["blk_t"], st["blk_o"], st["nblocks"], st["partial"], st["ctl"],
                     st["step_dev"], scaler_t, flags, acc[4:5], float(cfg.grad_clip), float(group["lr"]), float(beta1),
                     float(beta2), float(group["eps"]), sp)
                self.model._hip_engine.notify_weights_updated()
                self.optimizer._opt_called = True          # the scheduler's "step() before optimizer.step()" check
                batches += 1
                del host
        if batches == 0:
            return [0.0, 0.0, 0.0], 0
        host = torch.cat([acc, st["step_dev"], flags.float()]).cpu().tolist()       # the epoch's one read-back
        for q in st["params"]:
            self.optimizer.state[q]["step"].fill_(host[5])
        if scaler_t is not None:
            self.scaler._scale.copy_(scaler_t[0])
            self.scaler._growth_tracker.copy_(scaler_t[1].to(torch.int32))
        if host[7]:
            raise ValueError(f"policy_target outside [0, {A}) reached the loss kernel")
        return host[:3], batches
