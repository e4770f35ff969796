"""Device-resident mirror of the reference's `shogi_gym.VecEnv`.

Reference: shogi-engine/crates/shogi-gym/src/vec_env.rs:556-855 (the PyO3 class), step_result.rs:31-97 (result types).
Same constructor, `reset()` / `step(actions)`, result attributes, counters and error behaviour -- but the N games live in
HBM and a step is two launches of `shogi_env.hip` (C ABI `ka_shogi_env_*`, include/keisei_amd.h).  Both observation modes
("default" 46 planes, "katago" 50) and both action modes ("default" 13 527 actions, "spatial" 11 259) exist; the KataGo loop
asks for katago + spatial (katago_loop.py:580-585).  `DefaultActionMapper` / `SpatialActionMapper` are the index arithmetic of
action_mapper.rs / spatial_action_mapper.rs on the host.

`output="numpy"` (default) returns host arrays like the reference does; `output="torch"` returns the device tensors
themselves -- every per-step field of a result (observations, masks, rewards, flags, players, metadata) alternates
between two buffers, so a StepResult stays intact until the step after the next one; keep `.clone()`s of what must live
longer.  `terminal_observations` is the reference's ONE persistent buffer (vec_env.rs:246): the row of a game that ended
stays until that game ends again.  This is the form `select_actions` and the device rollout store consume without a host round trip.
A refused step (an illegal action anywhere) moves nothing: the kernel then writes the unchanged positions' observations
and masks (zero rewards, no flags) into the buffers the caller flips to, so a caller that runs with `check_actions=False`
and reads the flag late (`raise_if_refused()`) has still stepped against the right masks.
There is no CPU fallback: without the HIP library or a GPU the constructor raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np
import torch

from . import _lib

ACTION_SPACE = 81 * 139            # spatial
DEFAULT_ACTION_SPACE = 81 * 80 * 2 + 81 * 7
OBS_CHANNELS = 50                  # katago
DEFAULT_OBS_CHANNELS = 46
MASK_WORDS = (ACTION_SPACE + 31) // 32
_DIRS = ((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1))     # spatial_action_mapper.rs:31-40


def _sq(v: int, what: str) -> int:
    if not 0 <= v < 81:
        raise ValueError(f"invalid square index: {v}")
    return v


class SpatialActionMapper:
    """spatial_action_mapper.rs:138-356: flat index = square * 139 + move type, in the mover's perspective."""

    action_space_size = ACTION_SPACE

    def encode_board_move(self, from_sq: int, to_sq: int, promote: bool, is_white: bool) -> int:
        f, t = _sq(from_sq, "from"), _sq(to_sq, "to")
        if f == t:
            raise ValueError("from_sq and to_sq must be different")
        if is_white:
            f, t = 80 - f, 80 - t
        dr, dc = t // 9 - f // 9, t % 9 - f % 9
        if dr == 0 or dc == 0 or abs(dr) == abs(dc):
            unit = ((dr > 0) - (dr < 0), (dc > 0) - (dc < 0))
            return f * 139 + (64 if promote else 0) + _DIRS.index(unit) * 8 + max(abs(dr), abs(dc)) - 1
        if abs(dr) == 2 and abs(dc) == 1:
            same = (dr > 0) == (dc > 0)
            return f * 139 + 128 + (0 if same else 1) * 2 + (1 if promote else 0)
        raise ValueError(f"Cannot encode board move from ({f // 9},{f % 9}) to ({t // 9},{t % 9}) — not a valid direction, "
                         "distance, or knight move")

    def encode_drop_move(self, to_sq: int, piece_type_idx: int, is_white: bool) -> int:
        t = _sq(to_sq, "to")
        if not 0 <= piece_type_idx < 7:
            raise ValueError(f"piece_type_idx {piece_type_idx} out of range (max 6)")
        return (80 - t if is_white else t) * 139 + 132 + piece_type_idx

    def decode(self, idx: int, is_white: bool) -> dict:
        if not 0 <= idx < ACTION_SPACE:
            raise ValueError(f"action index {idx} out of range (max {ACTION_SPACE - 1})")
        s, slot = divmod(idx, 139)
        flip = (lambda q: 80 - q) if is_white else (lambda q: q)
        if slot >= 132:
            return {"type": "drop", "to_sq": flip(s), "piece_type_idx": slot - 132}
        if slot < 128:
            promote, b = slot >= 64, slot & 63
            dr, dc = _DIRS[b // 8]
            r, c = s // 9 + dr * (b % 8 + 1), s % 9 + dc * (b % 8 + 1)
        else:
            promote, (r, c) = bool((slot - 128) & 1), (s // 9 - 2, s % 9 + (-1 if (slot - 128) // 2 == 0 else 1))
        if not (0 <= r < 9 and 0 <= c < 9):
            raise ValueError(f"decoded move goes off board: from ({s // 9},{s % 9}) slot={slot}")
        return {"type": "board", "from_sq": flip(s), "to_sq": flip(r * 9 + c), "promote": promote}


class DefaultActionMapper:
    """action_mapper.rs:17-222: from * 160 + (to skipping from) * 2 + promote, then 81 x 7 drops."""

    action_space_size = DEFAULT_ACTION_SPACE

    def encode_board_move(self, from_sq: int, to_sq: int, promote: bool, is_white: bool) -> int:
        f, t = _sq(from_sq, "from"), _sq(to_sq, "to")
        if f == t:
            raise ValueError("from_sq and to_sq must be different")
        if is_white:
            f, t = 80 - f, 80 - t
        return f * 160 + (t - 1 if t > f else t) * 2 + (1 if promote else 0)

    def encode_drop_move(self, to_sq: int, piece_type_idx: int, is_white: bool) -> int:
        t = _sq(to_sq, "to")
        if not 0 <= piece_type_idx < 7:
            raise ValueError(f"piece_type_idx {piece_type_idx} is out of range (max 6)")
        return 81 * 160 + (80 - t if is_white else t) * 7 + piece_type_idx

    def decode(self, idx: int, is_white: bool) -> dict:
        if not 0 <= idx < DEFAULT_ACTION_SPACE:
            raise ValueError(f"action index {idx} is out of range (max {DEFAULT_ACTION_SPACE - 1})")
        flip = (lambda q: 80 - q) if is_white else (lambda q: q)
        if idx >= 81 * 160:
            t, h = divmod(idx - 81 * 160, 7)
            return {"type": "drop", "to_sq": flip(t), "piece_type_idx": h}
        f, rem = divmod(idx, 160)
        off = rem // 2
        return {"type": "board", "from_sq": flip(f), "to_sq": flip(off + 1 if off >= f else off), "promote": bool(rem & 1)}

_SFEN = {1: "P", 2: "L", 3: "N", 4: "S", 5: "G", 6: "B", 7: "R", 8: "K"}


@dataclass
class StepMetadata:          # step_result.rs:31-47
    captured_piece: Any
    termination_reason: Any
    ply_count: Any
    material_balance: Any


@dataclass
class StepResult:            # step_result.rs:50-83
    observations: Any
    legal_masks: Any
    rewards: Any
    terminated: Any
    truncated: Any
    terminal_observations: Any
    current_players: Any
    step_metadata: StepMetadata
    legal_mask_bits: Any = None      # (N, 352) int32 packed rows (torch output only): the device rollout store's column


@dataclass
class ResetResult:           # step_result.rs:86-97
    observations: Any
    legal_masks: Any
    legal_mask_bits: Any = None


class VecEnv:
    def __init__(self, num_envs: int = 512, max_ply: int = 500, observation_mode: str = "default",
                 action_mode: str = "default", *, device: Optional[torch.device] = None, output: str = "numpy",
                 check_actions: bool = True):
        if observation_mode not in ("default", "katago"):
            raise ValueError(f"Unknown observation_mode '{observation_mode}'. Valid: 'default', 'katago'")
        if action_mode not in ("default", "spatial"):
            raise ValueError(f"Unknown action_mode '{action_mode}'. Valid: 'default', 'spatial'")
        self._omode, self._amode = int(observation_mode == "katago"), int(action_mode == "spatial")
        self._A = ACTION_SPACE if self._amode else DEFAULT_ACTION_SPACE
        self._C = OBS_CHANNELS if self._omode else DEFAULT_OBS_CHANNELS
        if output not in ("numpy", "torch"):
            raise ValueError("output must be 'numpy' or 'torch'")
        if num_envs <= 0 or max_ply < 0 or max_ply > 65535:
            raise ValueError("num_envs must be positive and 0 <= max_ply <= 65535")
        _lib._load()                                          # raises KeiseiHipError when the library is missing
        if not torch.cuda.is_available():
            raise _lib.KeiseiHipError("keisei_amd.shogi_gym.VecEnv needs a GPU (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._n, self._max_ply, self._output, self._check = int(num_envs), int(max_ply), output, bool(check_actions)
        n, dev, hist = self._n, self.device, max(self._max_ply, 1)
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=dev)
        self._state = z(n, _lib.query("ka_shogi_env_state_bytes"), dtype=torch.uint8)
        self._keys = z(n, hist, dtype=torch.int64)
        self._checks = z(n, hist, dtype=torch.uint8)
        self._obs = [z(n, self._C, 9, 9, dtype=torch.float32) for _ in range(2)]
        self._mask = [z(n, self._A, dtype=torch.bool) for _ in range(2)]
        self._bits = [z(n, (self._A + 31) // 32, dtype=torch.int32) for _ in range(2)]
        self._cur = 0
        two = lambda *shape, dtype: [z(*shape, dtype=dtype) for _ in range(2)]
        self._rewards = two(n, dtype=torch.float32)
        self._terminated = two(n, dtype=torch.bool)
        self._truncated = two(n, dtype=torch.bool)
        # ONE persistent buffer, as the reference's terminal_obs_buffer (vec_env.rs:246,598): a game's row is rewritten only
        # when that game ends again, rows of running games keep what they held
        self._terminal_obs = z(n, self._C, 9, 9, dtype=torch.float32)
        self._players = two(n, dtype=torch.uint8)
        self._captured = [torch.full((n,), 255, dtype=torch.uint8, device=dev) for _ in range(2)]
        self._reason = two(n, dtype=torch.uint8)
        self._ply = two(n, dtype=torch.int16)                 # u16 payload (max_ply <= 65535); viewed as uint16 on the host
        self._material = two(n, dtype=torch.int32)
        self._stats = z(4, dtype=torch.int64)
        self._err = z(2, dtype=torch.int64)                   # [this step's refusal, the latch raise_if_refused reads and clears]
        self._actions = z(n, dtype=torch.int64)
        # as in the reference's constructor (vec_env.rs:574-612): the games stand at the start position, the mask buffer
        # is still all-false -- a step() before reset() is refused ("action index ... is not legal")
        self.reset()
        for t in (self._obs[0], self._mask[0], self._bits[0]):
            t.zero_()
        self._armed = False

    # ------------------------------------------------------------------ core
    def reset(self) -> ResetResult:
        """vec_env.rs:617-645: every game back to the start position; observations and masks of the first move."""
        self._armed = True
        with torch.cuda.device(self.device):
            self._cur = 0
            self._err.zero_()                                 # (a refusal nobody asked about ends with the games it belonged to)
            _lib.call("ka_shogi_env_reset", self._state, self._keys, self._checks, self._n, self._max_ply, self._omode,
                      self._amode, self._obs[0], self._mask[0], self._bits[0], self._players[0], 0, _lib.stream_ptr())
        return ResetResult(self._out(self._obs[0]), self._out(self._mask[0]),
                           self._bits[0] if self._output == "torch" else None)

    def step(self, actions) -> StepResult:
        """vec_env.rs:651-790.  `actions`: N action indices (list, numpy array or tensor; a CUDA int64 tensor is used in place)."""
        n = self._n
        if isinstance(actions, torch.Tensor):
            if actions.numel() != n:
                raise ValueError(f"expected {n} actions, got {actions.numel()}")
            act = actions.reshape(n)
            if act.device != self.device or act.dtype != torch.int64 or not act.is_contiguous():
                self._actions.copy_(act.to(torch.int64), non_blocking=True)
                act = self._actions
        else:
            a = np.asarray(actions, dtype=np.int64).reshape(-1)
            if a.shape[0] != n:
                raise ValueError(f"expected {n} actions, got {a.shape[0]}")
            self._actions.copy_(torch.from_numpy(a), non_blocking=False)
            act = self._actions
        if not getattr(self, "_armed", True):                 # no masks were handed out yet: every action is refused
            a0 = int(act[0].item())
            if a0 < 0:
                raise ValueError(f"env 0: negative action index {a0}")
            raise RuntimeError(f"env 0: action index {a0} is not legal")
        prev, nxt = self._cur, self._cur ^ 1
        with torch.cuda.device(self.device):
            _lib.call("ka_shogi_env_step", self._state, self._keys, self._checks, act, n, self._max_ply, self._omode, self._amode,
                      self._mask[prev], self._bits[prev], self._err, self._obs[nxt], self._mask[nxt], self._bits[nxt],
                      self._rewards[nxt], self._terminated[nxt], self._truncated[nxt], self._terminal_obs, self._players[nxt],
                      self._captured[nxt], self._reason[nxt], self._ply[nxt], self._material[nxt], self._stats, _lib.stream_ptr())
        self._cur = nxt                                       # (a refused step has re-written the unchanged positions there)
        if self._check:
            self.raise_if_refused(act)
        o = self._out
        ply = self._ply[nxt] if self._output == "torch" else self._ply[nxt].cpu().numpy().view(np.uint16)
        meta = StepMetadata(o(self._captured[nxt]), o(self._reason[nxt]), ply, o(self._material[nxt]))
        return StepResult(o(self._obs[nxt]), o(self._mask[nxt]), o(self._rewards[nxt]), o(self._terminated[nxt]),
                          o(self._truncated[nxt]), o(self._terminal_obs), o(self._players[nxt]), meta,
                          self._bits[nxt] if self._output == "torch" else None)

    def raise_if_refused(self, actions: Optional[torch.Tensor] = None) -> None:
        """The reference refuses a step before anything moves (vec_env.rs:660-690); so does the kernel, and this reads
        its latch (one 8-byte copy).  With check_actions=False call it whenever convenient: a refused step moved no game,
        its result holds the unchanged positions again (zero rewards, no flags), and the latch keeps the FIRST refusal --
        env index and action, stored by the kernel -- through any number of later steps until it is reported here."""
        word = int(self._err[1].item())
        if word == 0:
            return
        self._err[1].zero_()
        i = self._n - (word >> 32)
        a = word & 0xFFFFFFFF
        a = a - (1 << 32) if a >= (1 << 31) else a
        if actions is not None and abs(a) >= (1 << 31) - 1:   # clamped by the kernel: the caller's own tensor has the exact index
            a = int(actions[i].item())
        if a < 0:
            raise ValueError(f"env {i}: negative action index {a}")
        raise RuntimeError(f"env {i}: action index {a} is not legal")

    def _out(self, t: torch.Tensor):
        return t if self._output == "torch" else t.cpu().numpy()

    # ------------------------------------------------------------------ properties (vec_env.rs:793-870)
    @property
    def action_space_size(self) -> int:
        return self._A

    @property
    def observation_channels(self) -> int:
        return self._C

    @property
    def num_envs(self) -> int:
        return self._n

    def _stat(self, i: int) -> int:
        return int(self._stats[i].item())

    @property
    def episodes_completed(self) -> int:
        return self._stat(0)

    @property
    def episodes_drawn(self) -> int:
        return self._stat(1)

    @property
    def episodes_truncated(self) -> int:
        return self._stat(2)

    @property
    def draw_rate(self) -> float:
        c = self._stat(0)
        return 0.0 if c == 0 else self._stat(1) / c

    @property
    def mean_episode_length(self) -> float:
        c = self._stat(0)
        return 0.0 if c == 0 else self._stat(3) / c

    @property
    def truncation_rate(self) -> float:
        c = self._stat(0)
        return 0.0 if c == 0 else self._stat(2) / c

    def reset_stats(self) -> None:
        self._stats.zero_()

    # ------------------------------------------------------------------ positions
    def get_state(self, game_id: int):
        """(board[81] piece bytes, hands[2][7], side to move, ply) of one game (piece bytes: piece.rs:10-19)."""
        if not 0 <= game_id < self._n:
            raise IndexError(f"game_id {game_id} out of range for {self._n} environments")
        raw = self._state[game_id].cpu().numpy()
        return raw[:81].copy(), raw[81:95].reshape(2, 7).copy(), int(raw[95]), int(raw[100:104].view(np.uint32)[0])

    def set_state(self, game_id: int, board, hands, side: int) -> None:
        """Place a position in one game (ply and history restart at 0) and refresh every game's observation and masks --
        the rule fixtures of the reference's tests build their positions square by square (rules.rs:575-1790)."""
        others = [self.get_state(i) for i in range(self._n) if i != game_id]
        if any(p != 0 for *_, p in others):
            raise RuntimeError("set_state refreshes all games from ply 0: call it right after reset()")
        raw = self._state.cpu().numpy()
        raw[game_id, :81] = np.asarray(board, np.uint8).reshape(81)
        raw[game_id, 81:95] = np.asarray(hands, np.uint8).reshape(14)
        raw[game_id, 95] = side
        self._refresh(raw)

    def set_states(self, boards, hands, sides) -> None:
        """All games at once: boards (N,81), hands (N,2,7) or (N,14), sides (N,); ply and history restart at 0."""
        raw = np.zeros(tuple(self._state.shape), np.uint8)
        raw[:, :81] = np.asarray(boards, np.uint8).reshape(self._n, 81)
        raw[:, 81:95] = np.asarray(hands, np.uint8).reshape(self._n, 14)
        raw[:, 95] = np.asarray(sides, np.uint8).reshape(self._n)
        self._refresh(raw)

    def _refresh(self, raw: np.ndarray) -> None:
        self._armed = True
        self._state.copy_(torch.from_numpy(raw))
        with torch.cuda.device(self.device):
            _lib.call("ka_shogi_env_reset", self._state, self._keys, self._checks, self._n, self._max_ply, self._omode,
                      self._amode, self._obs[self._cur], self._mask[self._cur], self._bits[self._cur], self._players[self._cur], 1,
                      _lib.stream_ptr())

    def current(self) -> ResetResult:
        """Observation and masks of the positions to move (what the last reset / step / set_state wrote)."""
        c = self._cur
        return ResetResult(self._out(self._obs[c]), self._out(self._mask[c]), self._bits[c] if self._output == "torch" else None)

    def get_sfen(self, game_id: int) -> str:
        """vec_env.rs:873-882 / sfen.rs:93-171."""
        board, hands, side, _ = self.get_state(game_id)
        rows = []
        for r in range(9):
            s, empty = "", 0
            for c in range(9):
                p = int(board[r * 9 + c])
                if not p:
                    empty += 1
                    continue
                if empty:
                    s, empty = s + str(empty), 0
                ch = _SFEN[p & 15]
                s += ("+" if p & 0x20 else "") + (ch.lower() if p & 0x10 else ch)
            rows.append(s + (str(empty) if empty else ""))
        hs = ""
        for color in (0, 1):
            for h in (6, 5, 4, 3, 2, 1, 0):                   # R B G S N L P
                cnt = int(hands[color, h])
                if cnt:
                    ch = _SFEN[h + 1]
                    hs += (str(cnt) if cnt > 1 else "") + (ch.lower() if color else ch)
        return f"{'/'.join(rows)} {'w' if side else 'b'} {hs or '-'} 1"

    def get_sfens(self) -> List[str]:
        return [self.get_sfen(i) for i in range(self._n)]

    def get_spectator_data(self):
        raise NotImplementedError("spectator dictionaries belong to the reference's web UI (out of scope, DESIGN.md §7)")
