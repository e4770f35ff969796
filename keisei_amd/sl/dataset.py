"""Binary position shards for supervised learning (API mirror of keisei/sl/dataset.py:16-208).

Shard file = a flat array of 16 220-byte records ``{obs f32[50*81], policy i64, value i64, score f32}``, no header; a
directory holds ``shard_<n>.bin`` files (numeric order) and an optional ``shard_meta.json`` whose ``placeholder: true``
marks all-zero pipeline-test data.  ``SLDataset`` keeps the reference's item interface (``__getitem__`` -> dict of
tensors, same validation messages) and adds ``read_batch(indices)``: one vectorised gather per shard through a
structured memory map, which is what the trainer's device path uses (a 4096-position batch is 66 MB; per-item
decoding through a DataLoader is ~100x slower than the GPU consumes positions).
"""
from __future__ import annotations

import bisect
import json
import logging
import re
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

logger = logging.getLogger(__name__)

OBS_SIZE = 50 * 81
OBS_BYTES = 4 * OBS_SIZE
RECORD_SIZE = OBS_BYTES + 8 + 8 + 4          # 16 220
SCORE_NORMALIZATION = 76.0                    # raw material difference -> ~[-1, 1]; shared with the RL buffer
NUM_ACTIONS = 11259

_RECORD = np.dtype([("obs", np.float32, (OBS_SIZE,)), ("policy", np.int64), ("value", np.int64), ("score", np.float32)])
assert _RECORD.itemsize == RECORD_SIZE
_SHARD_NAME = re.compile(r"shard_(\d+)\.bin$")


def write_shard(path: Path, observations: np.ndarray, policy_targets: np.ndarray, value_targets: np.ndarray,
                score_targets: np.ndarray) -> None:
    """Write n positions as one shard file (dataset.py:47-72)."""
    n = observations.shape[0]
    assert observations.shape == (n, OBS_SIZE)
    assert policy_targets.shape == (n,)
    assert value_targets.shape == (n,)
    assert score_targets.shape == (n,)
    records = np.empty(n, dtype=_RECORD)
    for field, column in (("obs", observations), ("policy", policy_targets), ("value", value_targets), ("score", score_targets)):
        records[field] = column.astype(_RECORD[field].base)
    records.tofile(path)


class SLDataset(Dataset):
    def __init__(self, data_dir: Path, max_cache_size: int = 16, allow_placeholder: bool = False) -> None:
        if max_cache_size < 1:
            raise ValueError(f"max_cache_size must be >= 1, got {max_cache_size}")
        meta_path = data_dir / "shard_meta.json"
        if meta_path.exists():
            try:
                meta = json.loads(meta_path.read_bytes())
            except json.JSONDecodeError as exc:
                raise ValueError(f"Corrupt shard_meta.json at {meta_path}: {exc}") from exc
            if meta.get("placeholder", False) and not allow_placeholder:
                raise ValueError(
                    f"Shard directory {data_dir} contains placeholder data "
                    f"(shard_meta.json has placeholder=true). These shards have "
                    f"all-zero observations and are not suitable for training. "
                    f"Pass allow_placeholder=True to override for pipeline testing.")
        self.data_dir = data_dir
        self.shards: list[tuple[Path, int]] = []
        self._cumulative: list[int] = []
        numbered = []
        for f in data_dir.glob("shard_*.bin"):
            m = _SHARD_NAME.search(f.name)
            numbered.append((int(m.group(1)) if m else -1, f))
        running = 0
        for _, f in sorted(numbered, key=lambda t: t[0]):
            size = f.stat().st_size
            if size % RECORD_SIZE:
                logger.warning("Shard %s has %d trailing bytes (file_size=%d, record_size=%d) "
                               "— possible corruption or interrupted write", f.name, size % RECORD_SIZE, size, RECORD_SIZE)
            count = size // RECORD_SIZE
            if count > 0:
                running += count
                self.shards.append((f, count))
                self._cumulative.append(running)
        self._total = running
        self._max_cache_size = max_cache_size
        self._mmap_cache: OrderedDict[Path, np.ndarray] = OrderedDict()
        if len(self.shards) > max_cache_size:
            logger.warning("SLDataset has %d shards but max_cache_size=%d; "
                           "consider increasing max_cache_size to reduce mmap re-opens", len(self.shards), max_cache_size)

    def __len__(self) -> int:
        return self._total

    def clear_cache(self) -> None:
        """Forget the open maps (DataLoader workers call this after fork)."""
        self._mmap_cache.clear()

    def _records(self, shard: int) -> np.ndarray:
        """Structured read-only view of one shard, LRU-cached."""
        path, count = self.shards[shard]
        view = self._mmap_cache.get(path)
        if view is None:
            view = np.memmap(path, dtype=_RECORD, mode="r", shape=(count,))
            self._mmap_cache[path] = view
            if len(self._mmap_cache) > self._max_cache_size:
                self._mmap_cache.popitem(last=False)
        else:
            self._mmap_cache.move_to_end(path)
        return view

    def _locate(self, idx: int) -> tuple[int, int]:
        shard = bisect.bisect_right(self._cumulative, idx)
        return shard, idx - (self._cumulative[shard - 1] if shard else 0)

    def _check_targets(self, policy: int, value: int, idx: int, shard: int, local: int) -> None:
        name = self.shards[shard][0].name
        if policy < 0 or policy >= NUM_ACTIONS:
            raise ValueError(f"Invalid policy_target={policy} at index {idx} "
                             f"(shard={name}, local={local}): must be in [0, 11259)")
        if value not in (0, 1, 2):
            raise ValueError(f"Invalid value_target={value} at index {idx} "
                             f"(shard={name}, local={local}): must be 0 (W), 1 (D), or 2 (L)")

    def __getitem__(self, idx: int) -> dict[str, torch.Tensor]:
        if idx < 0 or idx >= self._total:
            raise IndexError(f"index {idx} out of range for dataset with {self._total} positions")
        shard, local = self._locate(idx)
        rec = self._records(shard)[local]
        policy, value = int(rec["policy"]), int(rec["value"])
        self._check_targets(policy, value, idx, shard, local)
        return {"observation": torch.from_numpy(np.array(rec["obs"], dtype=np.float32).reshape(50, 9, 9)),
                "policy_target": torch.tensor(policy, dtype=torch.long),
                "value_target": torch.tensor(value, dtype=torch.long),
                "score_target": torch.tensor(float(rec["score"]), dtype=torch.float32)}

    def read_batch(self, indices, pin: bool = False) -> dict[str, torch.Tensor]:
        """The collated batch ``default_collate([self[i] for i in indices])`` would give, gathered shard by shard."""
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        n = idx.shape[0]
        if n and (idx.min() < 0 or idx.max() >= self._total):
            bad = int(idx[(idx < 0) | (idx >= self._total)][0])
            raise IndexError(f"index {bad} out of range for dataset with {self._total} positions")
        make = (lambda *shape, dtype: torch.empty(*shape, dtype=dtype).pin_memory()) if pin else \
            (lambda *shape, dtype: torch.empty(*shape, dtype=dtype))
        obs = make(n, 50, 9, 9, dtype=torch.float32)
        policy, value = make(n, dtype=torch.long), make(n, dtype=torch.long)
        score = make(n, dtype=torch.float32)
        obs_np, pol_np, val_np, sc_np = obs.numpy().reshape(n, OBS_SIZE), policy.numpy(), value.numpy(), score.numpy()
        shard_of = np.searchsorted(np.asarray(self._cumulative, dtype=np.int64), idx, side="right")
        for shard in np.unique(shard_of):
            rows = np.nonzero(shard_of == shard)[0]
            local = idx[rows] - (self._cumulative[shard - 1] if shard else 0)
            recs = self._records(int(shard))[local]          # fancy index on the map: one gather, a private copy
            obs_np[rows], pol_np[rows], val_np[rows], sc_np[rows] = recs["obs"], recs["policy"], recs["value"], recs["score"]
        bad_p = np.nonzero((pol_np < 0) | (pol_np >= NUM_ACTIONS))[0]
        bad_v = np.nonzero((val_np < 0) | (val_np > 2))[0]
        if bad_p.size or bad_v.size:
            first = int(min(bad_p[0] if bad_p.size else n, bad_v[0] if bad_v.size else n))
            shard, local = self._locate(int(idx[first]))
            self._check_targets(int(pol_np[first]), int(val_np[first]), int(idx[first]), shard, local)
        return {"observation": obs, "policy_target": policy, "value_target": value, "score_target": score}
