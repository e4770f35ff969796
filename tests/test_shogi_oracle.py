"""Pins oracle/shogi_oracle.c to the known answers the reference's own tests hold.  No Rust toolchain
exists here, so these fixtures -- positions and expected outcomes restated from shogi-core/src/{game,rules}.rs and
shogi-gym/src/{vec_env,observation,katago_observation,spatial_action_mapper}.rs test modules -- are the pin."""
import numpy as np
import pytest

from oracle import shogi as S
from oracle.shogi import BISHOP, GOLD, KING, LANCE, PAWN, ROOK, SILVER, WHITE, OracleVecEnv, sq


def _env(board, hands, side, max_ply=500):
    e = OracleVecEnv(1, max_ply)
    e.set_state(0, board, hands, side)
    return e


def test_perft_from_the_start_position():
    # game.rs:1225-1243 (30, 900, 25 470); 719 731 is the published depth-4 count for shogi
    e = OracleVecEnv(1)
    assert [e.perft(d) for d in (1, 2, 3)] == [30, 900, 25470]
    assert e.perft(4) == 719731


def test_start_position_mask_and_observation():
    # vec_env.rs:1083-1101, 1590-1626; observation.rs / katago_observation.rs start-position tests
    e = OracleVecEnv(3)
    obs, mask = e.reset()
    assert obs.shape == (3, 50, 9, 9) and mask.shape == (3, 11259)
    assert mask.sum(axis=1).tolist() == [30, 30, 30]
    o = obs[0]
    assert o[0].sum() == 9 and o[14].sum() == 9                      # nine pawns a side
    assert np.all(o[0][6] == 1) and np.all(o[14][2] == 1)            # own pawns on row 6, the opponent's on row 2
    assert o[7][8, 4] == 1 and o[21][0, 4] == 1                      # kings
    assert o[5][7, 1] == 1 and o[6][7, 7] == 1                       # own bishop / rook
    assert o[19][1, 7] == 1 and o[20][1, 1] == 1                     # the opponent's bishop / rook
    assert np.all(o[8:14] == 0) and np.all(o[22:28] == 0)            # nothing promoted
    assert np.all(o[28:42] == 0)                                     # empty hands
    assert np.all(o[42] == 1) and np.all(o[43] == 0)                 # Black to move, ply 0
    assert np.all(o[44:50] == 0)                                     # no repetition, no check, reserved plane
    assert not np.isnan(obs).any()


def test_observation_flips_for_white_and_counts_ply():
    e = OracleVecEnv(1, max_ply=100)
    _, mask = e.reset()
    a = S.encode(sq(6, 2), sq(5, 2))                                 # Black pushes the pawn on column 2
    assert mask[0, a]
    r = e.step([a])
    o = r["observations"][0]
    assert r["current_players"][0] == 1 and np.all(o[42] == 0)
    assert np.all(o[43] == np.float32(1) / np.float32(100))
    # seen from White the board is turned by 180 degrees: White's own pawns sit on row 6 again, and the pushed
    # Black pawn (5,2) appears at (3,6)
    assert np.all(o[0][6] == 1) and o[14][3, 6] == 1 and o[14][2, 6] == 0
    assert r["legal_masks"][0].sum() == 30
    assert r["step_metadata"]["ply_count"][0] == 1 if "step_metadata" in r else r["ply_count"][0] == 1
    assert r["material_balance"][0] == 0 and r["captured_piece"][0] == 255 and r["rewards"][0] == 0


def test_action_index_contract():
    # spatial_action_mapper.rs:376-389 (flat index = square * 139 + move type; N, distance 4 -> slot 3)
    assert S.encode(40, 40 - 4 * 9) == 40 * 139 + 3
    seen = set()
    for white in (False, True):
        for to in range(81):
            for h in range(1, 8):
                idx = S.encode(0, to, drop=h, white=white)
                assert 132 <= idx % 139 <= 138
                assert S.decode(idx, white) == (0, to, 0, h)
                if not white:
                    seen.add(idx)
    assert len(seen) == 81 * 7
    for d, (dr, dc) in enumerate(zip(S_DR, S_DC)):                  # sliding round trip from the centre, :417-443
        for dist in range(1, 9):
            r, c = 4 + dr * dist, 4 + dc * dist
            if not (0 <= r < 9 and 0 <= c < 9):
                continue
            for promote in (0, 1):
                idx = S.encode(40, sq(r, c), promote)
                assert idx == 40 * 139 + promote * 64 + d * 8 + dist - 1
                assert S.decode(idx) == (40, sq(r, c), promote, 0)
    for to, slot in ((sq(2, 3), 128), (sq(2, 5), 130)):              # knight jumps, :445-464
        for promote in (0, 1):
            idx = S.encode(40, to, promote)
            assert idx == 40 * 139 + slot + promote
            assert S.decode(idx) == (40, to, promote, 0)
    # perspectives, :466-494: the same move has different indices for the two colours and both round-trip
    ib, iw = S.encode(20, 11), S.encode(20, 11, white=True)
    assert ib != iw and S.decode(ib) == (20, 11, 0, 0) and S.decode(iw, True) == (20, 11, 0, 0)
    # a White knight moves down the board; seen from White it is the same forward jump
    assert S.encode(sq(0, 1), sq(2, 2), white=True) % 139 in (128, 130)
    assert S.decode(11259) is None
    assert S.decode(0 * 139 + 0) is None                             # N from row 0 leaves the board


S_DR = (-1, -1, 0, 1, 1, 1, 0, -1)
S_DC = (0, 1, 1, 1, 0, -1, -1, -1)


def _ufz_position(pinned_gold=False):
    b, h = S.empty_board()
    b[sq(0, 0)] = KING | WHITE
    b[sq(8, 8)] = KING
    b[sq(0, 8)] = ROOK
    b[sq(2, 1)] = GOLD
    b[sq(8, 0)] = LANCE
    if pinned_gold:
        b[sq(0, 1)] = GOLD | WHITE
    h[0, 0] = 1
    return b, h


def test_pawn_drop_mate_fixtures():
    # rules.rs:575-617 positive; :622-650 king escapes; :654-684 no check; :1356-1416 pinned defender;
    # :1420-1468 White drops; :1472-1504 the king takes the pawn
    e = _env(*_ufz_position(), 0)
    assert e.uchi_fu_zume(0, sq(1, 0), 0)
    _, mask = e.observe(0)
    assert not mask[S.encode(0, sq(1, 0), drop=PAWN)]                # and so the drop is not in the legal mask
    e = _env(*_ufz_position(pinned_gold=True), 0)
    assert e.uchi_fu_zume(0, sq(1, 0), 0)

    b, h = S.empty_board()
    b[sq(0, 4)] = KING | WHITE; b[sq(8, 4)] = KING; h[0, 0] = 1
    e = _env(b, h, 0)
    assert not e.uchi_fu_zume(0, sq(1, 4), 0)
    _, mask = e.observe(0)
    assert mask[S.encode(0, sq(1, 4), drop=PAWN)] and not mask[S.encode(0, sq(0, 3), drop=PAWN)]   # (last rank: dead drop)

    b, h = S.empty_board()
    b[sq(0, 0)] = KING | WHITE; b[sq(8, 8)] = KING; h[0, 0] = 1
    e = _env(b, h, 0)
    assert not e.uchi_fu_zume(0, sq(4, 4), 0)

    b, h = S.empty_board()
    b[sq(8, 8)] = KING; b[sq(0, 0)] = KING | WHITE; b[sq(8, 0)] = ROOK | WHITE; b[sq(6, 7)] = GOLD | WHITE
    b[sq(0, 8)] = LANCE | WHITE; h[1, 0] = 1
    e = _env(b, h, 1)
    assert e.uchi_fu_zume(0, sq(7, 8), 1)


def _two_kings():
    b, h = S.empty_board()
    b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE
    return b, h


def test_repetition_fixtures():
    # rules.rs:692-752 (fourfold by king shuttle = Repetition), :756-810 (threefold is not enough)
    cycle = ((sq(8, 4), sq(7, 4)), (sq(0, 4), sq(1, 4)), (sq(7, 4), sq(8, 4)), (sq(1, 4), sq(0, 4)))
    e = _env(*_two_kings(), 0)
    for _ in range(2):
        for f, t in cycle:
            e.play(0, f, t)
    assert e.sennichite() == (S.R_PROGRESS, -1) and e.repetition_count() == 3
    for f, t in cycle:
        e.play(0, f, t)
    assert e.sennichite() == (S.R_REPETITION, -1) and e.repetition_count() == 4
    obs, _ = e.observe(0)
    assert np.all(obs[46] == 1) and np.all(obs[44] == 0) and np.all(obs[45] == 0) and np.all(obs[47] == 0)


def test_perpetual_check_fixtures():
    # rules.rs:827-905: Black's rook chases the White king; White, the side being checked, wins.  :1508-1585 mirrored.
    b, h = S.empty_board()
    b[sq(0, 0)] = KING | WHITE; b[sq(8, 8)] = KING; b[sq(0, 8)] = ROOK
    e = _env(b, h, 1)
    assert e.in_check(0, 1)
    for _ in range(3):
        e.play(0, sq(0, 0), sq(1, 0)); e.play(0, sq(0, 8), sq(1, 8)); e.play(0, sq(1, 0), sq(0, 0)); e.play(0, sq(1, 8), sq(0, 8))
    assert e.sennichite() == (S.R_PERPETUAL, 1)

    b, h = S.empty_board()
    b[sq(8, 8)] = KING; b[sq(0, 0)] = KING | WHITE; b[sq(8, 0)] = ROOK | WHITE
    e = _env(b, h, 0)
    assert e.in_check(0, 0)
    for _ in range(3):
        e.play(0, sq(8, 8), sq(7, 8)); e.play(0, sq(8, 0), sq(7, 0)); e.play(0, sq(7, 8), sq(8, 8)); e.play(0, sq(7, 0), sq(8, 0))
    assert e.sennichite() == (S.R_PERPETUAL, 0)


def _impasse_position(black_pawns, black_rooks, white_pawns, white_rooks):
    # rules.rs:1190-1236 make_impasse_position
    b, h = S.empty_board()
    b[sq(0, 4)] = KING; b[sq(8, 4)] = KING | WHITE
    n = 0
    for r in range(3):
        for c in range(9):
            if (r, c) != (0, 4) and n < black_pawns:
                b[sq(r, c)] = PAWN; n += 1
    n = 0
    for r in range(6, 9):
        for c in range(9):
            if (r, c) != (8, 4) and n < white_pawns:
                b[sq(r, c)] = PAWN | WHITE; n += 1
    h[0, 6], h[1, 6] = black_rooks, white_rooks
    return b, h


def test_impasse_fixtures():
    # rules.rs:474-481 (start position: 27 points a side, nobody in the zone), :1254-1330
    e = OracleVecEnv(1)
    assert e.impasse_score(0, 0) == 27 and e.impasse_score(0, 1) == 27
    assert e.zone_count(0, 0) == 0 and e.zone_count(0, 1) == 0 and e.impasse() == (S.R_PROGRESS, -1)
    e = _env(*_impasse_position(9, 3, 9, 0), 0)
    assert e.zone_count(0, 0) == 10 and e.zone_count(0, 1) == 10 and e.impasse_score(0, 0) == 24 and e.impasse_score(0, 1) == 9
    assert e.impasse() == (S.R_IMPASSE, 0)
    assert _env(*_impasse_position(9, 0, 9, 3), 0).impasse() == (S.R_IMPASSE, 1)
    assert _env(*_impasse_position(9, 0, 9, 0), 0).impasse() == (S.R_PROGRESS, -1)
    assert _env(*_impasse_position(9, 3, 9, 3), 0).impasse() == (S.R_IMPASSE, -1)
    b, h = S.empty_board()                                           # :1588-1650 promoted pieces keep their base value
    b[sq(4, 4)] = PAWN | S.PROM; b[sq(4, 5)] = BISHOP | S.PROM; b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE
    e = _env(b, h, 0)
    assert e.impasse_score(0, 0) == 6


def test_material_balance_and_rewards():
    # rules.rs:957-1055; vec_env.rs:986-1060
    e = OracleVecEnv(1)
    assert e.material(0, 0) == 0 and e.material(0, 1) == 0
    b, h = S.empty_board()
    b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE; b[sq(4, 4)] = ROOK
    e = _env(b, h, 0)
    assert e.material(0, 0) == 10 and e.material(0, 1) == -10
    b[sq(4, 4)] = ROOK | S.PROM; h[1, 0] = 2; h[0, 3] = 1
    e = _env(b, h, 0)
    assert e.material(0, 0) == 12 - 2 + 5
    assert S.reward(S.R_CHECKMATE, 0, 0) == 1 and S.reward(S.R_CHECKMATE, 0, 1) == -1
    assert S.reward(S.R_PERPETUAL, 1, 1) == 1 and S.reward(S.R_PERPETUAL, 1, 0) == -1
    assert S.reward(S.R_IMPASSE, 0, 0) == 1 and S.reward(S.R_IMPASSE, 0, 1) == -1 and S.reward(S.R_IMPASSE, -1, 0) == 0
    for r in (S.R_REPETITION, S.R_MAXMOVES, S.R_PROGRESS):
        assert S.reward(r, -1, 0) == 0


def test_max_ply_truncation_auto_reset_and_counters():
    # vec_env.rs:1261-1378: two plies with max_ply = 2 truncate, the env restarts from the start position, the terminal
    # observation keeps the finished game, the counters move
    e = OracleVecEnv(2, max_ply=2)
    obs0, mask = e.reset()
    rng = np.random.default_rng(0)
    r = e.step([int(rng.choice(np.flatnonzero(m))) for m in mask])
    assert not r["truncated"].any() and not r["terminated"].any() and r["ply_count"].tolist() == [1, 1]
    r2 = e.step([int(rng.choice(np.flatnonzero(m))) for m in r["legal_masks"]])
    assert r2["truncated"].all() and not r2["terminated"].any()
    assert r2["termination_reason"].tolist() == [S.R_MAXMOVES] * 2 and r2["rewards"].tolist() == [0, 0]
    assert r2["ply_count"].tolist() == [2, 2]
    np.testing.assert_array_equal(r2["observations"], obs0)
    assert r2["legal_masks"].sum(axis=1).tolist() == [30, 30] and r2["current_players"].tolist() == [0, 0]
    assert np.all(r2["terminal_observations"][:, 43] == 1.0)         # ply / max_ply = 1 in the game that ended
    st = e.stats()
    assert st == dict(episodes_completed=2, episodes_drawn=0, episodes_truncated=2, total_episode_ply=4)


def test_an_illegal_action_is_refused_before_anything_moves():
    # vec_env.rs:651-690
    e = OracleVecEnv(2)
    _, mask = e.reset()
    good = int(np.flatnonzero(mask[0])[0])
    bad = int(np.flatnonzero(~mask[1])[0])
    before = e.state(0)
    with pytest.raises(RuntimeError, match="env 1"):
        e.step([good, bad])
    with pytest.raises(RuntimeError, match="env 0"):
        e.step([-1, good])
    after = e.state(0)
    assert all(np.array_equal(x, y) for x, y in zip(before, after))


def test_random_playouts_keep_the_invariants():
    """Every piece is somewhere (40 in all), masks are never empty while a game runs, mates score +1 for the mover."""
    e = OracleVecEnv(8, max_ply=200)
    _, mask = e.reset()
    rng = np.random.default_rng(1)
    ends = 0
    for _ in range(400):
        acts = [int(rng.choice(np.flatnonzero(m))) for m in mask]
        r = e.step(acts)
        mask = r["legal_masks"]
        assert mask.any(axis=1).all()
        for i in range(8):
            board, hands, side, ply = e.state(i)
            assert (board != 0).sum() + hands.sum() == 40
            assert side == r["current_players"][i]
        done = r["terminated"] | r["truncated"]
        ends += int(done.sum())
        mate = r["termination_reason"] == S.R_CHECKMATE
        assert np.all(r["rewards"][mate] == 1.0)
    assert ends > 0 and e.stats()["episodes_completed"] == ends


def test_default_modes_46_planes_and_13527_actions():
    # shogi-gym/tests/test_vec_env.py:124-146 (shapes, 30 legal moves) and action_mapper.rs:17-110 (index layout)
    e = OracleVecEnv(2, 100, "default", "default")
    obs, mask = e.reset()
    assert obs.shape == (2, 46, 9, 9) and mask.shape == (2, 13527) and mask.sum(axis=1).tolist() == [30, 30]
    k = OracleVecEnv(2, 100)
    kobs, kmask = k.reset()
    assert np.array_equal(obs[:, :44], kobs[:, :44]) and np.all(obs[:, 44:] == 0)     # katago = default + 6 planes
    # the same moves under both encodings
    for i in np.flatnonzero(kmask[0]):
        f, t, p, d = S.decode(int(i))
        assert mask[0, S.encode(f, t, p, d, spatial=False)]
    assert S.encode(0, 1, spatial=False) == 0 and S.encode(0, 80, True, spatial=False) == 79 * 2 + 1
    assert S.encode(5, 3, spatial=False) == 5 * 160 + 3 * 2 and S.encode(5, 7, spatial=False) == 5 * 160 + 6 * 2
    assert S.encode(0, 10, drop=3, spatial=False) == 12960 + 10 * 7 + 2
    assert S.encode(0, 10, drop=3, white=True, spatial=False) == 12960 + 70 * 7 + 2
    for idx in (0, 159, 160, 12959, 12960, 13526):
        f, t, p, d = S.decode(idx, spatial=False)
        assert S.encode(f, t, p, d, spatial=False) == idx
        fw, tw, pw, dw = S.decode(idx, white=True, spatial=False)
        assert S.encode(fw, tw, pw, dw, white=True, spatial=False) == idx
    assert S.decode(13527, spatial=False) is None
    # one truncated game: same bookkeeping in the default modes (test_vec_env.py:159-197, 257-273)
    e = OracleVecEnv(1, 1, "default", "default")
    _, mask = e.reset()
    r = e.step([int(np.flatnonzero(mask[0])[0])])
    assert r["truncated"][0] and r["legal_masks"][0].sum() == 30 and r["terminal_observations"].shape == (1, 46, 9, 9)
    assert r["terminal_observations"][0].sum() != 0 and r["current_players"][0] == 0
    assert e.stats()["episodes_completed"] == 1 and e.stats()["episodes_truncated"] == 1


def test_host_action_mappers_agree_with_the_oracle_and_the_env_needs_a_gpu():
    import torch

    from keisei_amd import _lib
    from keisei_amd.shogi_gym import DefaultActionMapper, SpatialActionMapper, VecEnv

    sp, df = SpatialActionMapper(), DefaultActionMapper()
    assert sp.action_space_size == 11259 and df.action_space_size == 13527
    for white in (False, True):
        for idx in range(0, 11259, 7):
            m = S.decode(idx, white)
            if m is None:
                with pytest.raises(ValueError):
                    sp.decode(idx, white)
            elif m[3]:
                assert sp.decode(idx, white) == {"type": "drop", "to_sq": m[1], "piece_type_idx": m[3] - 1}
                assert sp.encode_drop_move(m[1], m[3] - 1, white) == idx
            else:
                assert sp.decode(idx, white) == {"type": "board", "from_sq": m[0], "to_sq": m[1], "promote": bool(m[2])}
                assert sp.encode_board_move(m[0], m[1], bool(m[2]), white) == idx
        for idx in range(0, 13527, 11):
            m = S.decode(idx, white, spatial=False)
            if m[3]:
                assert df.decode(idx, white) == {"type": "drop", "to_sq": m[1], "piece_type_idx": m[3] - 1}
                assert df.encode_drop_move(m[1], m[3] - 1, white) == idx
            else:
                assert df.decode(idx, white) == {"type": "board", "from_sq": m[0], "to_sq": m[1], "promote": bool(m[2])}
                assert df.encode_board_move(m[0], m[1], bool(m[2]), white) == idx
    with pytest.raises(ValueError):
        sp.encode_board_move(3, 3, False, False)
    with pytest.raises(ValueError):
        sp.encode_board_move(0, 12, False, False)
    with pytest.raises(ValueError):
        df.decode(13527, False)
    with pytest.raises(ValueError, match="Unknown action_mode"):
        VecEnv(num_envs=2, action_mode="x")
    if not torch.cuda.is_available():                       # the product path fails loudly: there is no CPU env behind this class
        with pytest.raises(_lib.KeiseiHipError):
            VecEnv(num_envs=2, observation_mode="katago", action_mode="spatial")


def _replay_golden(make_env, step_fn):
    import hashlib
    from pathlib import Path

    g = np.load(Path(__file__).parent / "golden" / "g10_shogi_playout.npz")
    keys = ("observations", "legal_masks", "rewards", "terminated", "truncated", "terminal_observations", "current_players",
            "captured_piece", "termination_reason", "ply_count", "material_balance")
    env = make_env(int(g["n"]), int(g["max_ply"]))
    obs, mask = env.reset() if not hasattr(env, "num_envs") else (lambda r: (r.observations, r.legal_masks))(env.reset())
    assert hashlib.sha256(obs.tobytes() + mask.tobytes()).digest() == g["reset_digest"].tobytes()
    for t, acts in enumerate(g["actions"]):
        out = step_fn(env, acts)
        h = hashlib.sha256()
        for k in keys:
            h.update(np.ascontiguousarray(out[k]).tobytes())
        assert h.digest() == g["digests"][t].tobytes(), f"step {t}"
    return env, g["stats"].tolist()


def test_oracle_reproduces_the_committed_playout():
    """tests/golden/g10_shogi_playout.npz (oracle/make_shogi_golden.py): the oracle's behaviour, frozen."""
    env, stats = _replay_golden(lambda n, mp: OracleVecEnv(n, mp), lambda e, a: e.step(a))
    st = env.stats()
    assert [st["episodes_completed"], st["episodes_drawn"], st["episodes_truncated"], st["total_episode_ply"]] == stats


def _kings(bk, wk):
    b, h = S.empty_board()
    b[sq(*bk)] = KING; b[sq(*wk)] = KING | WHITE
    return b, h


def _legal(e, i=0):
    _, mask = e.observe(i)
    return [S.decode(int(a), white=bool(e.state(i)[2])) for a in np.flatnonzero(mask)]


def test_game_rs_known_answers():
    """Positions and expectations restated from shogi-core/src/game.rs's test module (line numbers of the reference)."""
    # :713-775 nifu: a pawn on column 4 forbids pawn drops there, other columns stay open
    b, h = _kings((8, 4), (0, 4)); b[sq(6, 4)] = PAWN; h[0, 0] = 1
    drops = [m for m in _legal(_env(b, h, 0)) if m[3] == PAWN]
    assert drops and all(m[1] % 9 != 4 for m in drops)
    # :1251-1286 a promoted pawn does not count for nifu
    b, h = _kings((8, 4), (0, 4)); b[sq(5, 4)] = PAWN | S.PROM; h[0, 0] = 1
    assert any(m[3] == PAWN and m[1] % 9 == 4 for m in _legal(_env(b, h, 0)))
    # :1288-1326 the same rule for White
    b, h = _kings((8, 4), (0, 4)); b[sq(3, 3)] = PAWN | WHITE; h[1, 0] = 1
    drops = [m for m in _legal(_env(b, h, 1)) if m[3] == PAWN]
    assert drops and all(m[1] % 9 != 3 for m in drops)
    # :806-850 mate: rook on the rank, gold beside the king, the gold protected
    b, h = _kings((0, 0), (8, 8)); b[sq(0, 8)] = ROOK | WHITE; b[sq(1, 1)] = GOLD | WHITE; b[sq(8, 1)] = ROOK | WHITE
    e = _env(b, h, 0)
    assert e.legal_count(0) == 0 and e.check_termination() == (S.R_CHECKMATE, 1)
    # :1061-1124 no legal move without check loses too
    b, h = _kings((0, 0), (2, 1)); b[sq(0, 1)] = b[sq(1, 0)] = b[sq(1, 1)] = PAWN | WHITE; b[sq(0, 5)] = ROOK | WHITE
    e = _env(b, h, 0)
    assert not e.in_check(0, 0) and e.legal_count(0) == 0 and e.check_termination() == (S.R_CHECKMATE, 1)
    # :2051-2110 and for White
    b, h = _kings((6, 7), (8, 8)); b[sq(8, 7)] = b[sq(7, 8)] = b[sq(7, 7)] = PAWN; b[sq(8, 3)] = ROOK
    e = _env(b, h, 1)
    if not e.in_check(0, 1) and e.legal_count(0) == 0:       # (mirror of the position above)
        assert e.check_termination() == (S.R_CHECKMATE, 0)
    # :902-960 a captured promoted piece goes to the hand as its base type
    b, h = _kings((8, 4), (0, 4)); b[sq(4, 4)] = BISHOP | S.PROM | WHITE; b[sq(4, 0)] = ROOK
    e = _env(b, h, 0)
    r = e.step([S.encode(sq(4, 0), sq(4, 4))])
    assert e.state(0)[1][0, BISHOP - 1] == 1 and r["captured_piece"][0] == BISHOP - 1 and e.state(0)[0][sq(4, 4)] == ROOK
    # :1441-1490 a pinned pawn has no move
    b, h = _kings((4, 4), (0, 0)); b[sq(4, 8)] = ROOK | WHITE; b[sq(4, 6)] = PAWN
    assert not [m for m in _legal(_env(b, h, 0)) if m[3] == 0 and m[0] == sq(4, 6)]
    # :1492-1544 in check: the king steps away or the bishop interposes, nothing else
    b, h = _kings((4, 4), (0, 0)); b[sq(4, 8)] = ROOK | WHITE; b[sq(6, 6)] = BISHOP
    e = _env(b, h, 0)
    moves = _legal(e)
    assert e.in_check(0, 0) and moves
    for f, t, p, d in moves:
        assert f == sq(4, 4) or (f == sq(6, 6) and t in (sq(4, 6), sq(4, 8))), (f, t)
    # :639-665 check detection; :623-637 thirty opening moves
    b, h = _kings((8, 4), (0, 4)); b[sq(4, 4)] = ROOK | WHITE
    assert _env(b, h, 0).in_check(0, 0)
    assert OracleVecEnv(1).legal_count(0) == 30


def _lone(piece, at, side=0, extra=()):
    b, h = S.empty_board()
    b[sq(*at)] = piece
    for pc, pos in extra:
        b[sq(*pos)] = pc
    e = _env(b, h, side)
    return e.pseudo_moves(0, side)


def test_movegen_rs_known_answers():
    """Pseudo-legal generator counts restated from shogi-core/src/movegen.rs's tests (:428-582, :697-828, :1056-1146)."""
    assert len(_lone(ROOK, (4, 4))) == 19                    # 16 targets, the three in the zone twice
    assert len(_lone(BISHOP, (4, 4))) == 22
    assert len(_lone(LANCE, (4, 4))) == 6                    # (3,4) once, (2,4) (1,4) twice, (0,4) promoted only
    m = _lone(PAWN | WHITE, (2, 4), side=1)
    assert m == [(sq(2, 4), sq(3, 4), 0, 0)]
    assert len(_lone(PAWN | WHITE, (5, 4), side=1)) == 2     # into the zone: with and without promotion
    assert _lone(PAWN | WHITE, (7, 4), side=1) == [(sq(7, 4), sq(8, 4), 1, 0)]
    m = _lone(S.KNIGHT | WHITE, (4, 4), side=1)
    assert len(m) == 4 and {t for _, t, _, _ in m} == {sq(6, 3), sq(6, 5)}
    m = _lone(S.KNIGHT, (4, 4))                              # :275-320 Black's knight jumps up the board
    assert {t for _, t, _, _ in m} == {sq(2, 3), sq(2, 5)} and len(m) == 4
    assert {t for _, t, _, _ in _lone(SILVER, (0, 0))} == {sq(1, 1)}
    assert {t for _, t, _, _ in _lone(SILVER, (0, 8))} == {sq(1, 7)}
    assert {t for _, t, _, _ in _lone(GOLD, (0, 0))} == {sq(0, 1), sq(1, 0)}
    assert len({t for _, t, _, _ in _lone(BISHOP | S.PROM, (0, 0))}) == 10
    assert len({t for _, t, _, _ in _lone(ROOK | S.PROM, (8, 8))}) == 17
    for pc in (PAWN, LANCE, S.KNIGHT, SILVER):               # :842-953 promoted minor pieces move like a gold
        assert {t for _, t, _, _ in _lone(pc | S.PROM, (4, 4))} == {t for _, t, _, _ in _lone(GOLD, (4, 4))}
    # :593-647 a slider stops before its own piece and does not take it
    m = _lone(ROOK, (4, 4), extra=((PAWN, (4, 6)),))
    assert sq(4, 5) in {t for _, t, _, _ in m} and sq(4, 6) not in {t for f, t, _, _ in m if f == sq(4, 4)}
    # :371-401, :1118-1146 dead drops: no pawn / lance on the last rank, no knight on the last two
    b, h = S.empty_board(); h[0] = [1, 1, 1, 1, 0, 0, 0]
    e = _env(b, h, 0)
    drops = [x for x in e.pseudo_moves(0, 0, boards_only=False) if x[3]]
    rows = lambda t: {x[1] // 9 for x in drops if x[3] == t}
    assert rows(PAWN) == set(range(1, 9)) and rows(LANCE) == set(range(1, 9)) and rows(S.KNIGHT) == set(range(2, 9))
    assert rows(SILVER) == set(range(9)) and len(drops) == 72 + 72 + 63 + 81
    # :242-273 / :501-519 twenty board moves... the opening has 30 pseudo-legal board moves a side
    e = OracleVecEnv(1)
    assert len(e.pseudo_moves(0, 0)) == 30 and len(e.pseudo_moves(0, 1)) == 30


def test_observation_rs_known_answers():
    """Plane contents restated from shogi-gym/src/observation.rs (:402-724) and katago_observation.rs (:214-420)."""
    b, h = _kings((8, 4), (0, 4)); h[0, 0] = 9; h[1, 6] = 1; h[1, 1] = 3
    o, _ = _env(b, h, 0).observe(0)
    assert np.all(o[28] == np.float32(9) / np.float32(18)) and np.all(o[29:35] == 0)         # own pawns in hand: 9 / 18
    assert np.all(o[35 + 6] == np.float32(0.5)) and np.all(o[35 + 1] == np.float32(0.75))    # the opponent's rook 1/2, lances 3/4
    o, _ = _env(b, h, 1).observe(0)                                                          # seen from White the hands swap
    assert np.all(o[35] == np.float32(0.5)) and np.all(o[28 + 6] == np.float32(0.5)) and np.all(o[42] == 0)
    b, h = _kings((8, 4), (0, 4)); b[sq(4, 4)] = BISHOP | S.PROM; b[sq(2, 2)] = ROOK | S.PROM | WHITE; b[sq(5, 1)] = PAWN | S.PROM
    o, _ = _env(b, h, 0).observe(0)
    assert o[12][4, 4] == 1 and o[5][4, 4] == 0                   # own horse: plane 8 + 4
    assert o[27][2, 2] == 1 and o[20][2, 2] == 0                  # the opponent's dragon: plane 22 + 5
    assert o[8][5, 1] == 1 and o[0][5, 1] == 0                    # own tokin: plane 8 + 0
    assert o[:28].sum() == 5 and np.all(o[:28].sum(axis=0) <= 1)  # one plane per occupied square
    o, _ = _env(b, h, 1).observe(0)                               # White's view: colours swap, the board turns
    assert o[22 + 4][4, 4] == 1 and o[13][6, 6] == 1 and o[22][3, 7] == 1
    e = _env(*_kings((8, 4), (0, 4)), 0, max_ply=0)               # :726 max_ply 0: the ply plane is 0, not NaN
    o, _ = e.observe(0)
    assert np.all(o[43] == 0) and not np.isnan(o).any()
    # repetition planes 44..47 after 1, 2, 3, 4 returns to a position, exactly one of them set
    cycle = ((sq(8, 4), sq(7, 4)), (sq(0, 4), sq(1, 4)), (sq(7, 4), sq(8, 4)), (sq(1, 4), sq(0, 4)))
    e = _env(*_kings((8, 4), (0, 4)), 0)
    for rep in range(1, 6):
        for f, t in cycle:
            e.play(0, f, t)
        o, _ = e.observe(0)
        want = 44 + min(rep, 4) - 1
        assert [bool(np.all(o[c] == 1)) for c in range(44, 48)] == [c == want for c in range(44, 48)], rep
    # check plane: set for the side to move whose king is attacked, from either perspective
    b, h = _kings((8, 4), (0, 4)); b[sq(4, 4)] = ROOK | WHITE
    assert np.all(_env(b, h, 0).observe(0)[0][48] == 1)
    b[sq(4, 4)] = ROOK
    assert np.all(_env(b, h, 1).observe(0)[0][48] == 1) and np.all(_env(b, h, 0).observe(0)[0][48] == 0)


def _amap(pieces, side=0):
    b, h = S.empty_board()
    for pc, pos in pieces:
        b[sq(*pos)] = pc
    return _env(b, h, side).attack_map(0).reshape(2, 9, 9)


def test_attack_rs_known_answers():
    """Attack counts restated from shogi-core/src/attack.rs's tests (:452-704)."""
    m = OracleVecEnv(1).attack_map(0).reshape(2, 9, 9)
    assert np.all(m[0][5] >= 1) and np.all(m[1][3] >= 1)                 # the pawn rows attack the rows in front of them
    m = _amap([(KING, (4, 4))])
    assert m[0].sum() == 8 and m[0][3:6, 3:6].sum() == 8 and m[0][4, 4] == 0
    m = _amap([(ROOK, (4, 4))])
    assert np.all(np.delete(m[0][4], 4) == 1) and np.all(np.delete(m[0][:, 4], 4) == 1) and m[0].sum() == 16
    m = _amap([(ROOK, (4, 4)), (PAWN | WHITE, (4, 6))])                  # the blocker's square is attacked, nothing behind it
    assert m[0][4, 5] == 1 and m[0][4, 6] == 1 and m[0][4, 7] == 0 and m[0][4, 8] == 0
    m = _amap([(S.KNIGHT, (4, 4))])
    assert m[0][2, 3] == 1 and m[0][2, 5] == 1 and m[0].sum() == 2
    m = _amap([(BISHOP | S.PROM, (4, 4))])
    assert m[0][3, 4] == m[0][5, 4] == m[0][4, 3] == m[0][4, 5] == 1 and m[0][3, 3] == 1 and m[0][0, 0] == 1
    m = _amap([(LANCE, (4, 4))])
    assert np.all(m[0][:4, 4] == 1) and np.all(m[0][5:, 4] == 0) and m[0].sum() == 4
    m = _amap([(ROOK, (4, 0)), (ROOK, (0, 4))])
    assert m[0][4, 4] == 2                                               # two attackers are counted twice
    m = _amap([(S.KNIGHT | WHITE, (4, 4)), (LANCE | WHITE, (2, 0))])
    assert m[1][6, 3] == 1 and m[1][6, 5] == 1 and np.all(m[1][3:, 0] == 1) and m[1][1, 0] == 0   # White's pieces point down the board


def test_vec_env_rs_behaviour_restated():
    """vec_env.rs:1400-1565, 1735-1783: players alternate, games are isolated from each other, the material balance is the
    last mover's view (positive right after it captures)."""
    e = OracleVecEnv(3, 500)
    obs, mask = e.reset()
    assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[1], obs[2])
    acts = [int(np.flatnonzero(m)[0]) for m in mask]
    acts[1] = int(np.flatnonzero(mask[1])[-1])
    r = e.step(acts)
    assert r["current_players"].tolist() == [1, 1, 1] and r["material_balance"].tolist() == [0, 0, 0]
    assert not np.array_equal(r["observations"][0], r["observations"][1]) and np.array_equal(r["observations"][0], r["observations"][2])
    b, h = _kings((8, 4), (0, 4)); b[sq(5, 4)] = PAWN; b[sq(4, 4)] = PAWN | WHITE; b[sq(3, 0)] = GOLD | WHITE
    e = _env(b, h, 0)
    r = e.step([S.encode(sq(5, 4), sq(4, 4))])                       # Black's pawn takes the pawn in front of it
    assert r["captured_piece"][0] == 0 and r["material_balance"][0] == (1 + 1) - 6      # board pawn + hand pawn against a gold
    assert e.state(0)[1][0, 0] == 1 and r["current_players"][0] == 1
    r = e.step([S.encode(sq(3, 0), sq(4, 0), white=True)])           # White's reply: now the balance is White's view
    assert r["material_balance"][0] == 6 - 2 and r["captured_piece"][0] == 255
