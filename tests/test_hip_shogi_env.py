"""SURVEY §8 f3 parity: the device VecEnv (shogi_env.hip behind keisei_amd.shogi_gym.VecEnv) against the CPU oracle
(oracle/shogi_oracle.c, pinned by tests/test_shogi_oracle.py) -- bit for bit, every output of every step."""
import numpy as np
import pytest
import torch

from keisei_amd import _lib

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from keisei_amd.shogi_gym import VecEnv  # noqa: E402
from oracle import shogi as S  # noqa: E402
from oracle.shogi import BISHOP, GOLD, KING, LANCE, PAWN, ROOK, SILVER, WHITE, OracleVecEnv, sq  # noqa: E402

FIELDS = ("observations", "legal_masks", "rewards", "terminated", "truncated", "terminal_observations", "current_players")
META = ("captured_piece", "termination_reason", "ply_count", "material_balance")


def _env(n, max_ply, observation_mode="katago", action_mode="spatial", **kw):
    return VecEnv(num_envs=n, max_ply=max_ply, observation_mode=observation_mode, action_mode=action_mode, **kw)


def _compare_step(dev, ref, tag):
    for k in FIELDS:
        a, b = getattr(dev, k), ref[k]
        assert a.dtype == b.dtype and a.shape == b.shape, (tag, k, a.dtype, b.dtype, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError(f"{tag}: {k} differs at {bad[:5].tolist()} ({len(bad)} entries)")
    for k in META:
        a, b = getattr(dev.step_metadata, k), ref[k]
        assert a.dtype == b.dtype and np.array_equal(a, b), (tag, k, a[:8], b[:8])


def _playout(n, max_ply, steps, seed, compare_states_every=25, modes=("katago", "spatial")):
    dev, ref = _env(n, max_ply, *modes), OracleVecEnv(n, max_ply, *modes)
    r0, (obs, mask) = dev.reset(), ref.reset()
    assert np.array_equal(r0.observations, obs) and np.array_equal(r0.legal_masks, mask)
    rng = np.random.default_rng(seed)
    reasons = np.zeros(6, int)
    for s in range(steps):
        acts = np.array([rng.choice(np.flatnonzero(m)) for m in mask], dtype=np.int64)
        rd, rr = dev.step(acts), ref.step(acts)
        _compare_step(rd, rr, f"step {s}")
        mask = rr["legal_masks"]
        done = rr["terminated"] | rr["truncated"]
        for x in rr["termination_reason"][done]:
            reasons[x] += 1
        if s % compare_states_every == 0:
            for i in range(0, n, max(1, n // 8)):
                bd, hd, sd, pd = dev.get_state(i)
                bo, ho, so, po = ref.state(i)
                assert np.array_equal(bd, bo) and np.array_equal(hd, ho) and (sd, pd) == (so, po)
    st = ref.stats()
    assert (dev.episodes_completed, dev.episodes_drawn, dev.episodes_truncated) == (
        st["episodes_completed"], st["episodes_drawn"], st["episodes_truncated"])
    assert dev.mean_episode_length == (st["total_episode_ply"] / st["episodes_completed"] if st["episodes_completed"] else 0.0)
    return reasons


def test_reset_matches_the_oracle_and_the_reference_facts():
    dev = _env(5, 500)
    r = dev.reset()
    obs, mask = OracleVecEnv(5, 500).reset()
    assert r.observations.dtype == np.float32 and r.observations.shape == (5, 50, 9, 9)
    assert r.legal_masks.dtype == np.bool_ and r.legal_masks.shape == (5, 11259)
    assert np.array_equal(r.observations, obs) and np.array_equal(r.legal_masks, mask)
    assert r.legal_masks.sum(axis=1).tolist() == [30] * 5                         # vec_env.rs:1083-1101
    assert dev.action_space_size == 11259 and dev.observation_channels == 50 and dev.num_envs == 5
    assert dev.get_sfen(0) == "lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL b - 1"


def test_random_playouts_short_games():
    """max_ply 40: many truncations and restarts; every output of 300 steps x 64 games is identical."""
    reasons = _playout(64, 40, 300, seed=1)
    assert reasons[S.R_MAXMOVES] > 100


def test_random_playouts_long_games():
    """max_ply 300: captures, drops, promotions, mates (random play mates in roughly a tenth of the games)."""
    reasons = _playout(96, 300, 700, seed=2)
    assert reasons[S.R_CHECKMATE] > 0


def test_default_modes_and_the_mixed_ones():
    """The reference's constructor defaults (46 planes, 13 527 actions: vec_env.rs:559-573) and the two mixed settings."""
    for k, modes in enumerate((("default", "default"), ("katago", "default"), ("default", "spatial"))):
        reasons = _playout(24, 60, 150, seed=10 + k, modes=modes)
        assert reasons[S.R_MAXMOVES] > 0
    env = VecEnv(num_envs=4, max_ply=100)                               # shogi-gym/tests/test_vec_env.py:124-146
    assert (env.num_envs, env.action_space_size, env.observation_channels) == (4, 13527, 46)
    r = env.reset()
    assert r.observations.shape == (4, 46, 9, 9) and r.legal_masks.shape == (4, 13527) and r.legal_masks.sum(1).tolist() == [30] * 4


def test_action_mappers_on_the_host():
    from keisei_amd.shogi_gym import DefaultActionMapper, SpatialActionMapper
    sp, df = SpatialActionMapper(), DefaultActionMapper()
    assert sp.action_space_size == 11259 and df.action_space_size == 13527
    rng = np.random.default_rng(3)
    for _ in range(300):
        white = bool(rng.integers(2))
        idx = int(rng.integers(11259))
        m = S.decode(idx, white)
        if m is None:
            with pytest.raises(ValueError):
                sp.decode(idx, white)
        else:
            d = sp.decode(idx, white)
            if m[3]:
                assert d == {"type": "drop", "to_sq": m[1], "piece_type_idx": m[3] - 1}
                assert sp.encode_drop_move(m[1], m[3] - 1, white) == idx
            else:
                assert d == {"type": "board", "from_sq": m[0], "to_sq": m[1], "promote": bool(m[2])}
                assert sp.encode_board_move(m[0], m[1], bool(m[2]), white) == idx
        idx = int(rng.integers(13527))
        m = S.decode(idx, white, spatial=False)
        d = df.decode(idx, white)
        if m[3]:
            assert d == {"type": "drop", "to_sq": m[1], "piece_type_idx": m[3] - 1} and df.encode_drop_move(m[1], m[3] - 1, white) == idx
        else:
            assert d == {"type": "board", "from_sq": m[0], "to_sq": m[1], "promote": bool(m[2])}
            assert df.encode_board_move(m[0], m[1], bool(m[2]), white) == idx
    with pytest.raises(ValueError):
        sp.encode_board_move(3, 3, False, False)
    with pytest.raises(ValueError):
        sp.encode_board_move(0, 12, False, False)                      # neither a line nor a knight jump
    with pytest.raises(ValueError):
        df.decode(13527, False)


def test_odd_env_counts_and_mask_alignment():
    # rows of 11 259 bytes start at every alignment: 1, 3 and 17 games cover head / body / tail of the byte writer
    for n in (1, 3, 17):
        _playout(n, 30, 45, seed=n)


def test_packed_masks_are_the_bool_masks():
    dev = _env(9, 50, output="torch")
    r = dev.reset()
    for _ in range(20):
        m = r.legal_masks
        bits = r.legal_mask_bits.cpu().numpy().view(np.uint32)
        ref = np.zeros((9, 352 * 32), bool); ref[:, :11259] = m.cpu().numpy()
        assert np.array_equal(np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little").astype(bool), ref)
        probs = m.float()
        acts = torch.multinomial(probs, 1).squeeze(1)
        r = dev.step(acts)
    assert r.observations.is_cuda and r.step_metadata.material_balance.dtype == torch.int32


def _fixture(board, hands, side, max_ply=500):
    dev, ref = _env(1, max_ply), OracleVecEnv(1, max_ply)
    dev.reset(); ref.reset()
    dev.set_state(0, board, hands, side); ref.set_state(0, board, hands, side)
    return dev, ref


def _same_view(dev, ref):
    cur = dev.current()
    obs, mask = ref.observe(0)
    assert np.array_equal(cur.observations[0], obs), np.argwhere(cur.observations[0] != obs)[:5]
    if not np.array_equal(cur.legal_masks[0], mask):
        d = np.flatnonzero(cur.legal_masks[0] != mask)
        raise AssertionError(f"mask differs at {[(int(i) // 139, int(i) % 139) for i in d[:8]]}")
    return mask


def _ufz(pinned=False):
    b, h = S.empty_board()
    b[sq(0, 0)] = KING | WHITE; b[sq(8, 8)] = KING; b[sq(0, 8)] = ROOK; b[sq(2, 1)] = GOLD; b[sq(8, 0)] = LANCE
    if pinned:
        b[sq(0, 1)] = GOLD | WHITE
    h[0, 0] = 1
    return b, h


def test_rule_fixture_positions_give_the_same_masks():
    """The reference's pawn-drop-mate / pin / escape fixtures (rules.rs:575-684, 1356-1504), plus a few positions with
    checks, pins, promotions zones and full hands."""
    cases = [(*_ufz(), 0), (*_ufz(True), 0)]
    b, h = S.empty_board(); b[sq(0, 4)] = KING | WHITE; b[sq(8, 4)] = KING; h[0, 0] = 1
    cases.append((b, h, 0))
    b, h = S.empty_board()
    b[sq(8, 8)] = KING; b[sq(0, 0)] = KING | WHITE; b[sq(8, 0)] = ROOK | WHITE; b[sq(6, 7)] = GOLD | WHITE; b[sq(0, 8)] = LANCE | WHITE
    h[1, 0] = 1
    cases.append((b, h, 1))
    # Black in check from a rook with a pinned silver and pieces to interpose from hand
    b, h = S.empty_board()
    b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE; b[sq(3, 4)] = ROOK | WHITE; b[sq(7, 3)] = SILVER; b[sq(5, 1)] = BISHOP | WHITE
    b[sq(6, 0)] = PAWN; b[sq(2, 2)] = PAWN | WHITE
    h[0] = [2, 1, 1, 1, 1, 1, 0]; h[1] = [3, 0, 2, 0, 1, 0, 1]
    cases.append((b, h, 0)); cases.append((b, h, 1))
    # promotion zones, must-promote ranks, knights on the edge, promoted pieces of both colours
    b, h = S.empty_board()
    b[sq(8, 0)] = KING; b[sq(0, 8)] = KING | WHITE
    b[sq(1, 2)] = PAWN; b[sq(2, 3)] = LANCE; b[sq(2, 0)] = S.KNIGHT; b[sq(3, 8)] = S.KNIGHT; b[sq(3, 5)] = SILVER; b[sq(2, 6)] = BISHOP
    b[sq(4, 4)] = ROOK | S.PROM; b[sq(5, 5)] = BISHOP | S.PROM | WHITE; b[sq(7, 6)] = PAWN | WHITE; b[sq(6, 3)] = LANCE | WHITE
    b[sq(6, 8)] = S.KNIGHT | WHITE; b[sq(5, 0)] = S.KNIGHT | WHITE; b[sq(6, 1)] = SILVER | S.PROM | WHITE; b[sq(1, 7)] = GOLD | WHITE
    h[0] = [1, 1, 1, 0, 0, 0, 0]; h[1] = [1, 1, 1, 1, 0, 0, 0]
    cases.append((b, h, 0)); cases.append((b, h, 1))
    for board, hands, side in cases:
        dev, ref = _fixture(board, hands, side)
        mask = _same_view(dev, ref)
        # and one step from there with every legal action in turn (at most 24 of them): same outputs
        for a in np.flatnonzero(mask)[:24]:
            d2, r2 = _fixture(board, hands, side)
            _compare_step(d2.step([int(a)]), r2.step([int(a)]), f"action {a}")


def _drive(dev, ref, moves, white_first=False):
    out = None
    for k, (f, t) in enumerate(moves):
        white = (k % 2 == 1) != white_first
        a = S.encode(f, t, white=white)
        rd, rr = dev.step([a]), ref.step([a])
        _compare_step(rd, rr, f"move {k}")
        out = rr
    return out


def test_repetition_and_perpetual_check_through_the_step_api():
    # rules.rs:692-752: king shuttle, fourth occurrence = Repetition (a draw: reward 0, counted as drawn)
    b, h = S.empty_board(); b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE
    cycle = [(sq(8, 4), sq(7, 4)), (sq(0, 4), sq(1, 4)), (sq(7, 4), sq(8, 4)), (sq(1, 4), sq(0, 4))]
    dev, ref = _fixture(b, h, 0)
    last = _drive(dev, ref, cycle * 3)
    assert last["termination_reason"][0] == S.R_REPETITION and last["terminated"][0] and last["rewards"][0] == 0
    assert dev.episodes_drawn == 1 and np.all(last["terminal_observations"][0, 46] == 1)
    # rules.rs:827-905: the rook chases the king; the checked side (White) wins, and White made the last move
    b, h = S.empty_board(); b[sq(0, 0)] = KING | WHITE; b[sq(8, 8)] = KING; b[sq(0, 8)] = ROOK
    dev, ref = _fixture(b, h, 1)
    chase = [(sq(0, 0), sq(1, 0)), (sq(0, 8), sq(1, 8)), (sq(1, 0), sq(0, 0)), (sq(1, 8), sq(0, 8))]
    last = _drive(dev, ref, chase * 3, white_first=True)
    assert last["termination_reason"][0] == S.R_PERPETUAL and last["rewards"][0] == -1.0      # Black moved last and loses


def test_repetition_heavy_play_matches_the_oracle():
    """The device env recognises a repeated position by a 64-bit key of (board, hands, side), the oracle by comparing whole
    positions: 96 sparse games (two kings and up to four other pieces, hands part of the position) in which both sides mostly
    take their last move back, with random other moves, drops and captures mixed in -- hundreds of fourfold repetitions and
    perpetual checks reached through transpositions, every output of every step identical."""
    n, rng = 96, np.random.default_rng(77)
    boards, hands, sides = np.zeros((n, 81), np.uint8), np.zeros((n, 2, 7), np.uint8), np.zeros(n, np.uint8)
    kinds = [GOLD, SILVER, ROOK, BISHOP, LANCE, S.KNIGHT, GOLD, SILVER]
    probe = OracleVecEnv(1, 300); probe.reset()
    i = 0
    while i < n:
        b, h = S.empty_board()
        free = list(rng.permutation(81))
        b[free.pop()] = KING; b[free.pop()] = KING | WHITE
        for _ in range(int(rng.integers(0, 5))):
            k = kinds[int(rng.integers(len(kinds)))]
            cell = free.pop()
            if k in (LANCE, S.KNIGHT) and not 2 <= cell // 9 <= 6: continue        # (no piece without a move)
            b[cell] = k | (WHITE if rng.random() < 0.5 else 0)
        if rng.random() < 0.5: h[int(rng.integers(2)), int(rng.integers(1, 7))] = 1
        side = int(rng.integers(2))
        probe.set_state(0, b, h, side)
        other_king = int(np.flatnonzero(b == (KING if side else KING | WHITE))[0])
        if probe.attack_map(0)[side][other_king] or probe.legal_count(0) == 0: continue   # (a legal position with a move to make)
        boards[i], hands[i], sides[i] = b, h, side
        i += 1
    dev, ref = _env(n, 300), OracleVecEnv(n, 300)
    dev.reset(); ref.reset()
    dev.set_states(boards, hands, sides)
    for i in range(n): ref.set_state(i, boards[i], hands[i], int(sides[i]))
    cur = dev.current()
    mask, players = cur.legal_masks, sides.copy()
    for i in range(n):
        assert np.array_equal(mask[i], ref.observe(i)[1]), i
    last = {}                                                # (game, colour) -> (from, to) of that side's previous board move
    reasons = np.zeros(6, int)
    for step in range(260):
        acts = np.zeros(n, np.int64)
        for i in range(n):
            w = bool(players[i])
            legal = np.flatnonzero(mask[i])
            a = None
            if (i, w) in last and rng.random() < 0.85:
                f, t = last[(i, w)]
                back = S.encode(t, f, white=w)
                if back >= 0 and mask[i][back]: a = back
            if a is None: a = int(rng.choice(legal))
            d = S.decode(a, white=w)                         # (from, to, promote, drop)
            if d is not None and d[3] == 0: last[(i, w)] = (d[0], d[1])
            else: last.pop((i, w), None)
            acts[i] = a
        rd, rr = dev.step(acts), ref.step(acts)
        _compare_step(rd, rr, f"step {step}")
        mask, players = rr["legal_masks"], rr["current_players"]
        done = rr["terminated"] | rr["truncated"]
        for i in np.flatnonzero(done):
            reasons[rr["termination_reason"][i]] += 1
            last.pop((i, False), None); last.pop((i, True), None)
    assert reasons[S.R_REPETITION] >= 100, reasons
    assert dev.episodes_drawn == ref.stats()["episodes_drawn"]


def test_impasse_through_the_step_api():
    # rules.rs:1190-1290: both kings entered, ten pieces each in the zone, Black reaches 24 points with three rooks in hand
    b, h = S.empty_board()
    b[sq(0, 4)] = KING; b[sq(8, 4)] = KING | WHITE
    n = 0
    for r in range(3):
        for c in range(9):
            if (r, c) != (0, 4) and n < 9:
                b[sq(r, c)] = PAWN | S.PROM; n += 1
    n = 0
    for r in range(6, 9):
        for c in range(9):
            if (r, c) != (8, 4) and n < 9:
                b[sq(r, c)] = PAWN | S.PROM | WHITE; n += 1
    h[0, 6] = 3
    b[sq(4, 0)] = GOLD                                                  # a quiet move to make
    dev, ref = _fixture(b, h, 0)
    a = S.encode(sq(4, 0), sq(3, 0))
    rd, rr = dev.step([a]), ref.step([a])
    _compare_step(rd, rr, "impasse")
    assert rr["termination_reason"][0] == S.R_IMPASSE and rr["rewards"][0] == 1.0


def _random_positions(count, seed):
    """Scattered pieces with both kings, no dead pieces, no doubled pawns, the side not to move not in check (the one thing
    legal play guarantees that the generators rely on); pins, multiple checks, full hands and crowded zones all occur."""
    rng = np.random.default_rng(seed)
    probe = OracleVecEnv(1, 500)
    out = []
    stock = {PAWN: 18, LANCE: 4, S.KNIGHT: 4, SILVER: 4, GOLD: 4, BISHOP: 2, ROOK: 2}
    while len(out) < count:
        b, h = S.empty_board()
        ks = rng.choice(81, 2, replace=False)
        b[ks[0]] = KING; b[ks[1]] = KING | WHITE
        left = dict(stock)
        pawn_cols = [set(), set()]
        for _ in range(int(rng.integers(4, 34))):
            t = int(rng.choice(list(left)))
            if left[t] == 0:
                continue
            sqr = int(rng.integers(81))
            if b[sqr]:
                continue
            color, prom = int(rng.integers(2)), bool(rng.integers(3) == 0) and t != GOLD
            row = sqr // 9
            if not prom:
                last = row == (8 if color else 0)
                last2 = row >= 7 if color else row <= 1
                if (t in (PAWN, LANCE) and last) or (t == S.KNIGHT and last2):
                    continue
                if t == PAWN:
                    if sqr % 9 in pawn_cols[color]:
                        continue
                    pawn_cols[color].add(sqr % 9)
            b[sqr] = t | (WHITE if color else 0) | (S.PROM if prom else 0)
            left[t] -= 1
        for t in left:
            k = int(rng.integers(0, left[t] + 1)) if rng.integers(2) else 0
            kb = int(rng.integers(0, k + 1))
            h[0, t - 1], h[1, t - 1] = kb, k - kb
        side = int(rng.integers(2))
        probe.set_state(0, b, h, side)
        if probe.in_check(0, side ^ 1) or probe.legal_count(0) == 0:
            continue
        out.append((b, h, side))
    return out


def test_random_positions_masks_observations_and_one_step():
    n = 384
    pos = _random_positions(n, seed=5)
    dev, ref = _env(n, 500), OracleVecEnv(n, 500)
    dev.reset(); ref.reset()
    dev.set_states([p[0] for p in pos], [p[1] for p in pos], [p[2] for p in pos])
    for i, (b, h, sd) in enumerate(pos):
        ref.set_state(i, b, h, sd)
    cur = dev.current()
    rng = np.random.default_rng(6)
    acts, checks = np.zeros(n, np.int64), 0
    for i in range(n):
        obs, mask = ref.observe(i)
        assert np.array_equal(cur.observations[i], obs), (i, np.argwhere(cur.observations[i] != obs)[:4])
        if not np.array_equal(cur.legal_masks[i], mask):
            d = np.flatnonzero(cur.legal_masks[i] != mask)
            raise AssertionError(f"position {i} ({dev.get_sfen(i)}): mask differs at {[(int(j) // 139, int(j) % 139) for j in d[:8]]}")
        checks += int(obs[48, 0, 0])
        acts[i] = rng.choice(np.flatnonzero(mask))
    assert checks > 10                                   # plenty of positions with the mover in check
    _compare_step(dev.step(acts), ref.step(acts), "step from random positions")


def test_refused_actions_raise_like_the_reference_and_move_nothing():
    dev = _env(4, 100)
    r = dev.reset()
    good = [int(np.flatnonzero(m)[0]) for m in r.legal_masks]
    bad = int(np.flatnonzero(~r.legal_masks[2])[0])
    before = [dev.get_state(i) for i in range(4)]
    with pytest.raises(RuntimeError, match=r"env 2: action index \d+ is not legal"):
        dev.step(good[:2] + [bad] + good[3:])
    with pytest.raises(ValueError, match="env 1: negative action index -5"):
        dev.step([good[0], -5, good[2], good[3]])
    with pytest.raises(RuntimeError, match="env 0"):
        dev.step([11259] + good[1:])
    with pytest.raises(ValueError, match="expected 4 actions, got 3"):
        dev.step(good[:3])
    after = [dev.get_state(i) for i in range(4)]
    for x, y in zip(before, after):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2:] == y[2:]
    r2 = dev.step(good)                                                 # and the env still works
    assert r2.step_metadata.ply_count.tolist() == [1, 1, 1, 1]
    fresh = _env(2, 100)                                                # vec_env.rs:574-612: no masks before reset()
    assert fresh.get_sfen(1).startswith("lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL b - 1")
    with pytest.raises(RuntimeError, match="env 0: action index 0 is not legal"):
        fresh.step([0, 0])
    with pytest.raises(RuntimeError, match="env 0: action index 0 is not legal"):
        fresh.step([0, 0])                                              # ... and a refused step hands none out either
    with pytest.raises(ValueError, match="Unknown observation_mode"):
        VecEnv(num_envs=2, observation_mode="x")


def test_a_refused_step_without_host_checks_leaves_every_buffer_consistent():
    """check_actions=False (the device loop reads the flag late): a step with an illegal action anywhere moves NO game
    (vec_env.rs:651-690), and the result it returns is the unchanged positions again -- observations, masks, players, ply --
    with zero rewards and no flags, so the caller's next actions are validated against the right masks.  Every per-step
    field alternates between two buffers: the result of step t is still intact after step t+1."""
    n = 6
    dev = _env(n, 60, output="torch", check_actions=False)
    ref = OracleVecEnv(n, 60)
    r0, (obs, mask) = dev.reset(), ref.reset()
    rng = np.random.default_rng(5)
    pick = lambda m: np.array([rng.choice(np.flatnonzero(row)) for row in m], dtype=np.int64)
    acts = pick(mask)
    r1 = dev.step(torch.from_numpy(acts).cuda()); o1 = ref.step(acts)
    keep = {k: getattr(r1, k).clone() for k in ("observations", "legal_masks", "rewards", "terminated", "truncated", "current_players")}
    keep_ply = r1.step_metadata.ply_count.clone()
    bad = pick(o1["legal_masks"])
    bad[3] = int(np.flatnonzero(~o1["legal_masks"][3])[0])
    before = dev._state.clone()
    r2 = dev.step(torch.from_numpy(bad).cuda())                         # refused: not raised here (no host check)
    assert torch.equal(dev._state, before)                              # nothing moved, in any game
    for k, v in keep.items():                                           # step 1's result is still intact ...
        assert torch.equal(getattr(r1, k), v), k
    assert torch.equal(r2.observations, r1.observations) and torch.equal(r2.legal_masks, r1.legal_masks)   # ... and step 2 re-states it
    assert torch.equal(r2.legal_mask_bits, r1.legal_mask_bits) and torch.equal(r2.current_players, r1.current_players)
    assert torch.equal(r2.step_metadata.ply_count, keep_ply)
    assert not bool(r2.terminated.any()) and not bool(r2.truncated.any()) and not bool((r2.rewards != 0).any())
    with pytest.raises(RuntimeError, match=r"env 3: action index \d+ is not legal"):
        dev.raise_if_refused()
    good = pick(o1["legal_masks"])                                       # the env goes on from the unchanged positions
    r3 = dev.step(torch.from_numpy(good).cuda()); o3 = ref.step(good)
    dev.raise_if_refused()
    assert np.array_equal(r3.observations.cpu().numpy(), o3["observations"]) and np.array_equal(r3.legal_masks.cpu().numpy(), o3["legal_masks"])
    assert np.array_equal(r3.rewards.cpu().numpy(), o3["rewards"]) and np.array_equal(r3.current_players.cpu().numpy(), o3["current_players"])


def test_a_refusal_read_late_survives_later_steps():
    """check_actions=False: the refusal is latched by the kernel (include/keisei_amd.h, ka_shogi_env_step: word 1 of `err`), so an
    illegal step followed by legal ones is still reported -- with the env index and the action of the FIRST refused step, also
    when the actions arrived as a device tensor -- and reporting clears the latch."""
    n = 5
    dev = _env(n, 60, output="torch", check_actions=False)
    ref = OracleVecEnv(n, 60)
    dev.reset(); _, mask = ref.reset()
    rng = np.random.default_rng(11)
    pick = lambda m: np.array([rng.choice(np.flatnonzero(row)) for row in m], dtype=np.int64)
    bad = pick(mask)
    bad[2] = int(np.flatnonzero(~mask[2])[-1])
    bad[4] = int(np.flatnonzero(~mask[4])[0])                            # a second offender: the lower env index is reported
    dev.step(torch.from_numpy(bad).cuda())                               # refused, nothing raised
    for _ in range(3):                                                   # legal steps from the unchanged positions
        acts = pick(mask)
        r = dev.step(torch.from_numpy(acts).cuda()); o = ref.step(acts)
        mask = o["legal_masks"]
        assert np.array_equal(r.legal_masks.cpu().numpy(), mask)
    later = pick(mask); later[1] = int(np.flatnonzero(~mask[1])[0])
    dev.step(torch.from_numpy(later).cuda())                             # a second refused step does not displace the first
    with pytest.raises(RuntimeError, match=rf"env 2: action index {bad[2]} is not legal"):
        dev.raise_if_refused()
    dev.raise_if_refused()                                               # reported once: the latch is clear again
    neg = pick(mask); neg[0] = -7
    dev.step(torch.from_numpy(neg).cuda())
    with pytest.raises(ValueError, match=r"env 0: negative action index -7"):
        dev.raise_if_refused()


def test_full_size_invariants_on_the_device():
    """4096 games x 300 steps with on-device sampling (no oracle at this size): 40 pieces in every game, never an empty
    mask, planes agree with the stored boards, finished games restart at the start position, counters add up."""
    n, steps = 4096, 300
    dev = _env(n, 120, output="torch", check_actions=False)
    r = dev.reset()
    start_obs = r.observations[0].clone()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    ends = 0
    for s in range(steps):
        acts = torch.multinomial(r.legal_masks.float(), 1, generator=g).squeeze(1)
        r = dev.step(acts)
        assert bool(r.legal_masks.any(dim=1).all())
        done = r.terminated | r.truncated
        ends += int(done.sum())
        if s % 50 == 49:
            dev.raise_if_refused()
            st = dev._state
            pieces = (st[:, :81] != 0).sum(1) + st[:, 81:95].sum(1)
            assert bool((pieces == 40).all())
            occ = r.observations[:, :28].sum(dim=1).reshape(n, 81)     # one plane per piece: occupancy, turned for White
            board_occ = (st[:, :81] != 0).float()
            white = st[:, 95].bool()
            board_occ[white] = board_occ[white].flip(1)
            assert torch.equal(occ, board_occ)
            if bool(done.any()):
                assert torch.equal(r.observations[done][0], start_obs)
                assert bool((r.current_players[done] == 0).all())
            mate = r.step_metadata.termination_reason == S.R_CHECKMATE
            assert bool((r.rewards[mate] == 1).all())
    assert dev.episodes_completed == ends and ends > n


def test_device_env_reproduces_the_committed_playout():
    """The same frozen playout (tests/golden/g10_shogi_playout.npz) through the HIP kernels: digest for digest."""
    from tests.test_shogi_oracle import _replay_golden

    def step(env, acts):
        r = env.step(acts)
        d = {k: getattr(r, k) for k in FIELDS}
        d.update({k: getattr(r.step_metadata, k) for k in META})
        return d

    env, stats = _replay_golden(lambda n, mp: _env(n, mp), step)
    assert [env.episodes_completed, env.episodes_drawn, env.episodes_truncated] == stats[:3]


def test_reference_vector_positions_on_the_device_env():
    """The positions of tests/test_shogi_reference_vectors.py (known answers restated from rules.rs / game.rs / katago_observation.rs
    with their reference lines) placed in the DEVICE env: same observation and mask as the oracle, and the reference's own numbers
    read off the device's outputs -- legal-move counts from the mask, the check plane, the material balance and the reward of the
    stalemate step from the step metadata."""
    from tests.test_shogi_reference_vectors import POSITIONS, position
    for name, (pieces, hands, side, want) in sorted(POSITIONS.items()):
        kinds = {p & 15 | (p & WHITE) for _, _, p in pieces}
        if KING not in kinds or (KING | WHITE) not in kinds:
            continue                                                     # one-king score positions: no game to play
        board, hnd, sd = position(pieces, hands, side)
        dev, ref = _fixture(board, hnd, sd)
        mask = _same_view(dev, ref)
        cur = dev.current()
        if "legal" in want:
            assert int(cur.legal_masks[0].sum()) == want["legal"], name
        if "in_check" in want:
            assert bool(np.all(cur.observations[0][48] == (1.0 if want["in_check"] else 0.0))), name
        if "material" in want and mask.any():
            # rules.rs:968-1055 through vec_env.rs:371-374: after the mover's move the balance is reported for the mover
            a = int(np.flatnonzero(mask)[0])
            rd, rr = dev.step([a]), ref.step(np.array([a]))
            assert int(rd.step_metadata.material_balance[0]) == int(rr["material_balance"][0])
            m = S.decode(a, white=bool(sd))
            if not m[3] and not board[m[1]]:                             # a quiet move: the balance is the table's, seen by the mover
                assert int(rd.step_metadata.material_balance[0]) == (want["material"] if sd == 0 else -want["material"]), name
    # game.rs:2051-2110 through a step: the rook arrives, White has no move and is not in check -> Checkmate{winner: Black}
    pieces, hands, _, _ = POSITIONS["stalemate_white"]
    pieces = [p for p in pieces if p[2] != ROOK] + [(0, 3, ROOK)]
    dev, ref = _fixture(*position(pieces, hands, 0))
    r = dev.step([S.encode(sq(0, 3), sq(8, 3))])
    assert bool(r.terminated[0]) and int(r.step_metadata.termination_reason[0]) == S.R_CHECKMATE and float(r.rewards[0]) == 1.0


def test_python_api_tests_of_the_reference_restated():
    """shogi-gym/tests/test_vec_env.py, the classes that do not need the spectator dictionaries, restated on the device env with the
    constructor defaults the reference's tests use (46 planes, 13 527 actions); the reference line beside each block."""
    first = lambda masks: [int(np.flatnonzero(m)[0]) for m in masks]
    # TestVecEnvSfen: test_vec_env.py:73-80, :82-87, :89-94, :96-102, :104-112
    env = VecEnv(num_envs=3, max_ply=100)
    r = env.reset()
    sfens = env.get_sfens()
    assert isinstance(sfens, list) and len(sfens) == 3 and all(isinstance(s, str) for s in sfens)
    assert sfens[0] == sfens[1] and "lnsgkgsnl" in sfens[0].lower()
    assert [env.get_sfen(i) for i in range(3)] == sfens
    for bad in (3, 100):
        with pytest.raises(IndexError):
            env.get_sfen(bad)
    before = env.get_sfen(0)
    env.step(first(r.legal_masks))
    assert env.get_sfen(0) != before
    # TestVecEnvEpisodeStats: :149-152, :154-157, :159-166, :168-174, :176-184, :186-196
    env = VecEnv(num_envs=1, max_ply=100)
    env.reset()
    assert env.mean_episode_length == 0.0 and env.truncation_rate == 0.0
    env = VecEnv(num_envs=2, max_ply=1)
    r = env.reset()
    env.step(first(r.legal_masks))
    assert env.episodes_completed == 2 and env.truncation_rate == 1.0 and env.mean_episode_length == 1.0
    env = VecEnv(num_envs=1, max_ply=1)
    r = env.reset()
    for _ in range(5):
        r = env.step(first(r.legal_masks))
    assert env.episodes_completed == 5 and env.mean_episode_length == 1.0
    env.reset_stats()
    assert env.mean_episode_length == 0.0 and env.truncation_rate == 0.0
    # TestVecEnvStepping: :199-216 shapes, :218-222 wrong count, :224-231 illegal action, :233-243 metadata
    env = VecEnv(num_envs=2, max_ply=100)
    r = env.reset()
    s = env.step(first(r.legal_masks))
    assert s.observations.shape == (2, 46, 9, 9) and s.legal_masks.shape == (2, 13527)
    assert s.rewards.shape == s.terminated.shape == s.truncated.shape == (2,)
    m = s.step_metadata
    assert m.captured_piece.shape == m.termination_reason.shape == m.ply_count.shape == (2,)
    # :275-284 terminal observations, :286-298 current players (White after the first move)
    assert s.terminal_observations.shape == (2, 46, 9, 9) and s.terminal_observations.dtype == np.float32
    assert s.current_players.shape == (2,) and s.current_players.dtype == np.uint8 and s.current_players.tolist() == [1, 1]
    with pytest.raises(ValueError):
        env.step([0])
    env1 = VecEnv(num_envs=1, max_ply=100)
    r1 = env1.reset()
    with pytest.raises(RuntimeError):
        env1.step([int(np.flatnonzero(~r1.legal_masks[0])[0])])
    # :245-255 twenty random steps of four games, :300-316 rewards are -1, 0 or 1 over thirty steps
    env = VecEnv(num_envs=4, max_ply=50)
    masks = env.reset().legal_masks
    rng = np.random.default_rng(5)
    for _ in range(30):
        s = env.step([int(rng.choice(np.flatnonzero(mk))) for mk in masks])
        assert set(np.unique(s.rewards).tolist()) <= {-1.0, 0.0, 1.0}
        masks = s.legal_masks
    # :257-273 auto-reset after a truncation: start-position masks (30 moves), non-zero terminal observation, Black to move
    env = VecEnv(num_envs=1, max_ply=1)
    s = env.step(first(env.reset().legal_masks))
    assert bool(s.truncated[0]) and int(s.legal_masks[0].sum()) == 30
    assert s.terminal_observations.shape == (1, 46, 9, 9) and float(s.terminal_observations[0].sum()) != 0.0 and int(s.current_players[0]) == 0
    # TestVecEnvObservation: :320-334 different moves give different observations, :336-346 piece planes are binary,
    # :348-353 mask width = action space, :355-370 a second reset() gives the first one's state
    env = VecEnv(num_envs=2, max_ply=100)
    r = env.reset()
    obs1, masks1 = r.observations.copy(), r.legal_masks.copy()
    assert obs1.dtype == np.float32 and set(np.unique(obs1[0, :28]).tolist()) <= {0.0, 1.0}
    assert masks1.shape == (2, env.action_space_size)
    acts = [int(np.flatnonzero(masks1[0])[0]), int(np.flatnonzero(masks1[1])[-1])]
    assert acts[0] != acts[1]
    s = env.step(acts)
    assert not np.array_equal(s.observations[0], s.observations[1])
    r2 = env.reset()
    assert np.array_equal(r2.observations, obs1) and np.array_equal(r2.legal_masks, masks1)
