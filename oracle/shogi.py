"""TEST INFRASTRUCTURE: ctypes face of oracle/shogi_oracle.c (the CPU oracle of SURVEY §8 row f3).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  `build()` compiles the C
file with gcc into oracle/_build/ (git-ignored, travels to the GPU box with the snapshot)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "shogi_oracle.c"
LIB = HERE / "_build" / "libshogi_oracle.so"

A_SIZE = 81 * 139                  # spatial action space
A_DEFAULT = 81 * 80 * 2 + 81 * 7   # default action space
PAWN, LANCE, KNIGHT, SILVER, GOLD, BISHOP, ROOK, KING = range(1, 9)
WHITE, PROM = 0x10, 0x20
R_PROGRESS, R_CHECKMATE, R_REPETITION, R_PERPETUAL, R_IMPASSE, R_MAXMOVES = range(6)


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-fopenmp", "-shared", "-fPIC", "-o", str(LIB), str(SRC)], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.so_create.restype = C.c_void_p
        _lib.so_perft.restype = C.c_uint64
        _lib.so_reward.restype = C.c_float
        _lib.so_reward.argtypes = [C.c_int, C.c_int, C.c_int]
    return _lib


def piece_value(piece_type: int, promoted: bool) -> int:
    return lib().so_piece_value(piece_type, int(promoted))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class OracleVecEnv:
    """The reference's VecEnv(num_envs, max_ply, "katago", "spatial") restated on the CPU (vec_env.rs:556-855)."""

    def __init__(self, num_envs: int, max_ply: int = 500, observation_mode: str = "katago", action_mode: str = "spatial"):
        self.n, self.max_ply = num_envs, max_ply
        self.omode, self.amode = int(observation_mode == "katago"), int(action_mode == "spatial")
        self.C, self.A = (50 if self.omode else 46), (A_SIZE if self.amode else A_DEFAULT)
        self.h = C.c_void_p(lib().so_create(num_envs, max_ply, self.omode, self.amode))
        self.terminal_obs = np.zeros((num_envs, self.C, 9, 9), np.float32)  # persistent, like the reference's buffer

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_destroy(self.h)
            self.h = None

    def reset(self):
        obs = np.zeros((self.n, self.C, 9, 9), np.float32)
        mask = np.zeros((self.n, self.A), np.uint8)
        lib().so_reset(self.h, _p(obs), _p(mask))
        return obs, mask.astype(bool)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int64)
        assert a.shape == (self.n,)
        out = dict(
            observations=np.zeros((self.n, self.C, 9, 9), np.float32), legal_masks=np.zeros((self.n, self.A), np.uint8),
            rewards=np.zeros(self.n, np.float32), terminated=np.zeros(self.n, np.uint8), truncated=np.zeros(self.n, np.uint8),
            current_players=np.zeros(self.n, np.uint8), captured_piece=np.zeros(self.n, np.uint8),
            termination_reason=np.zeros(self.n, np.uint8), ply_count=np.zeros(self.n, np.uint16),
            material_balance=np.zeros(self.n, np.int32))
        rc = lib().so_step(self.h, _p(a), _p(out["observations"]), _p(out["legal_masks"]), _p(out["rewards"]),
                           _p(out["terminated"]), _p(out["truncated"]), _p(self.terminal_obs), _p(out["current_players"]),
                           _p(out["captured_piece"]), _p(out["termination_reason"]), _p(out["ply_count"]),
                           _p(out["material_balance"]))
        if rc != 0:
            raise RuntimeError(f"env {-rc - 1}: action index {int(a[-rc - 1])} is not legal")
        out["terminal_observations"] = self.terminal_obs.copy()
        for k in ("legal_masks", "terminated", "truncated"):
            out[k] = out[k].astype(bool)
        return out

    def stats(self):
        s = np.zeros(4, np.uint64)
        lib().so_stats(self.h, _p(s))
        return dict(zip(("episodes_completed", "episodes_drawn", "episodes_truncated", "total_episode_ply"), map(int, s)))

    def state(self, i: int):
        board, hands = np.zeros(81, np.uint8), np.zeros(14, np.uint8)
        side, ply = C.c_int(), C.c_int()
        lib().so_get_state(self.h, i, _p(board), _p(hands), C.byref(side), C.byref(ply))
        return board, hands.reshape(2, 7), side.value, ply.value

    def set_state(self, i: int, board, hands, side: int):
        b = np.ascontiguousarray(board, dtype=np.uint8).reshape(81)
        hd = np.ascontiguousarray(hands, dtype=np.uint8).reshape(14)
        lib().so_set_state(self.h, i, _p(b), _p(hd), int(side))

    def observe(self, i: int):
        obs, mask = np.zeros((self.C, 9, 9), np.float32), np.zeros(self.A, np.uint8)
        lib().so_observe(self.h, i, _p(obs), _p(mask))
        return obs, mask.astype(bool)

    # rule-level probes used by the fixtures
    def legal_count(self, i=0): return lib().so_legal_count(self.h, i)

    def attack_map(self, i=0):
        out = np.zeros((2, 81), np.uint8)
        lib().so_attack_map(self.h, i, _p(out))
        return out

    def pseudo_moves(self, i, color, boards_only=True):
        out = np.zeros(2048, np.uint32)
        n = lib().so_pseudo_moves(self.h, i, color, int(boards_only), _p(out))
        return [(int(v & 255), int((v >> 8) & 255), int((v >> 16) & 255), int(v >> 24)) for v in out[:n]]

    def in_check(self, i, color): return bool(lib().so_in_check(self.h, i, color))
    def uchi_fu_zume(self, i, to, color): return bool(lib().so_uchi_fu_zume(self.h, i, to, color))
    def impasse_score(self, i, color): return lib().so_impasse_score(self.h, i, color)
    def zone_count(self, i, color): return lib().so_zone_count(self.h, i, color)
    def material(self, i, who): return lib().so_material(self.h, i, who)
    def piece_attacks(self, i, frm, piece, target): return bool(lib().so_piece_attacks(self.h, i, frm, piece, target))
    def repetition_count(self, i=0): return lib().so_repetition_count(self.h, i)
    def perft(self, depth, i=0): return int(lib().so_perft(self.h, i, depth))
    def play(self, i, frm, to, promote=False, drop=0): lib().so_play(self.h, i, frm, to, int(promote), drop)

    def impasse(self, i=0):
        w = C.c_int()
        return lib().so_impasse(self.h, i, C.byref(w)), w.value

    def sennichite(self, i=0):
        w = C.c_int()
        return lib().so_sennichite(self.h, i, C.byref(w)), w.value

    def check_termination(self, i=0):
        w = C.c_int()
        return lib().so_check_termination(self.h, i, C.byref(w)), w.value


def encode(frm, to, promote=False, drop=0, white=False, spatial=True):
    return lib().so_encode(frm, to, int(promote), drop, int(white), int(spatial))


def decode(idx, white=False, spatial=True):
    out = (C.c_int * 4)()
    if lib().so_decode(idx, int(white), int(spatial), out) != 0:
        return None
    return tuple(out)


def reward(result, winner, last_mover):
    return float(lib().so_reward(result, winner, last_mover))


def empty_board():
    return np.zeros(81, np.uint8), np.zeros((2, 7), np.uint8)


def sq(row, col):
    return row * 9 + col
