"""torchrun environment -> process-group helpers (mirror of keisei/training/distributed.py:30-157).

One process per GPU; backend "nccl" (= RCCL on ROCm, over xGMI) when a GPU is visible, "gloo"
otherwise.  Rendezvous uses MASTER_ADDR/MASTER_PORT from the environment (127.0.0.1 on one node).
"""
from __future__ import annotations

import dataclasses
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


def _resolve_device(is_distributed: bool, local_rank: int) -> torch.device:
    if not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device(f"cuda:{local_rank}") if is_distributed else torch.device("cuda")


@dataclasses.dataclass(frozen=True, slots=True)
class DistributedContext:
    rank: int
    local_rank: int
    world_size: int
    is_distributed: bool
    device: torch.device = dataclasses.field(init=False)

    def __post_init__(self) -> None:
        object.__setattr__(self, "device", _resolve_device(self.is_distributed, self.local_rank))

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def _require_env(key: str) -> str:
    val = os.environ.get(key)
    if val is None:
        raise RuntimeError(
            f"torchrun env var {key!r} is missing. Ensure RANK, LOCAL_RANK, and WORLD_SIZE are all set. "
            f"Launch with: torchrun --nproc_per_node=N your_script.py")
    return val


def get_distributed_context() -> DistributedContext:
    rank = os.environ.get("RANK")
    if rank is None:
        return DistributedContext(rank=0, local_rank=0, world_size=1, is_distributed=False)
    return DistributedContext(rank=int(rank), local_rank=int(_require_env("LOCAL_RANK")),
                              world_size=int(_require_env("WORLD_SIZE")), is_distributed=True)


def setup_distributed(ctx: DistributedContext, backend: str | None = None) -> None:
    if not ctx.is_distributed:
        return
    have_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if have_gpu else "gloo"
    elif backend == "nccl" and not have_gpu:
        raise RuntimeError(
            "backend='nccl' requires CUDA but torch.cuda.is_available() is False. "
            "Use backend='gloo' for CPU-only distributed training, or set backend=None to auto-select.")
    try:
        if have_gpu:
            torch.cuda.set_device(ctx.local_rank)
        dist.init_process_group(backend=backend)
        logger.info("DDP initialized: rank=%d, local_rank=%d, world_size=%d, backend=%s",
                    ctx.rank, ctx.local_rank, ctx.world_size, backend)
    except Exception:
        logger.error("DDP init failed: rank=%d, local_rank=%d, world_size=%d, MASTER_ADDR=%s, MASTER_PORT=%s",
                     ctx.rank, ctx.local_rank, ctx.world_size, os.environ.get("MASTER_ADDR", "<unset>"),
                     os.environ.get("MASTER_PORT", "<unset>"))
        raise


def cleanup_distributed(ctx: DistributedContext) -> None:
    if ctx.is_distributed and dist.is_initialized():
        dist.destroy_process_group()
        logger.info("DDP process group destroyed: rank=%d", ctx.rank)


def seed_all_ranks(seed: int) -> None:
    """Seeds torch / numpy / random; call with base_seed + rank."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
