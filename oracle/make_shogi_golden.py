"""Writes tests/golden/g10_shogi_playout.npz: a seeded random playout of the env oracle (12 games, max_ply 60, 240 steps) as
the chosen actions plus a digest of every output of every step, and the final counters.  The fixture freezes the oracle's
behaviour (pinned separately to the reference's known answers by tests/test_shogi_oracle.py) so that later changes to the
oracle or to the device env cannot drift together unnoticed.  Run from the repository root: python oracle/make_shogi_golden.py"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.shogi import OracleVecEnv  # noqa: E402

KEYS = ("observations", "legal_masks", "rewards", "terminated", "truncated", "terminal_observations", "current_players",
        "captured_piece", "termination_reason", "ply_count", "material_balance")


def digest(step: dict) -> np.ndarray:
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(step[k]).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def main() -> None:
    n, max_ply, steps = 12, 60, 240
    env = OracleVecEnv(n, max_ply)
    obs, mask = env.reset()
    rng = np.random.default_rng(20260101)
    actions, digests = [], []
    h0 = hashlib.sha256(obs.tobytes() + mask.tobytes()).digest()
    for _ in range(steps):
        a = np.array([rng.choice(np.flatnonzero(m)) for m in mask], dtype=np.int64)
        r = env.step(a)
        actions.append(a); digests.append(digest(r))
        mask = r["legal_masks"]
    st = env.stats()
    np.savez_compressed("tests/golden/g10_shogi_playout.npz", n=n, max_ply=max_ply, actions=np.stack(actions),
                        digests=np.stack(digests), reset_digest=np.frombuffer(h0, dtype=np.uint8),
                        stats=np.array([st["episodes_completed"], st["episodes_drawn"], st["episodes_truncated"], st["total_episode_ply"]]))
    print("wrote tests/golden/g10_shogi_playout.npz", st)


if __name__ == "__main__":
    main()
