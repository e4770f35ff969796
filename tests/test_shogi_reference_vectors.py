"""More of the known answers the reference's own tests hold for SURVEY §8 f3, restated table by table with the reference line
beside each expectation (VERDICT r2 "What's missing" 5).  Every expectation here is a number, a mask or a plane the reference
asserts -- not an output of this repository's oracle: the oracle (oracle/shogi_oracle.c) and the host action mappers
(keisei_amd.shogi_gym) are what is being pinned.  tests/test_hip_shogi_env.py runs the position tables below through the
device env as well (`POSITIONS`, `MAPPER_CASES`).

Reference files (under /root/reference/shogi-engine/crates/): shogi-core/src/{rules,game}.rs, shogi-gym/src/{vec_env,
katago_observation,spatial_action_mapper,action_mapper}.rs, shogi-gym/tests/test_{vec_env,observation,action_mapper}.py.
Counts of reference tests restated per module: DESIGN.md §4a."""
import numpy as np
import pytest

from oracle import shogi as S
from oracle.shogi import BISHOP, GOLD, KING, KNIGHT, LANCE, PAWN, PROM, ROOK, SILVER, WHITE, OracleVecEnv, sq
from tests.test_shogi_oracle import _env

HAND = {PAWN: 0, LANCE: 1, KNIGHT: 2, SILVER: 3, GOLD: 4, BISHOP: 5, ROOK: 6}          # HandPieceType::ALL order (types.rs)


def position(pieces, hands=(), side=0):
    """pieces: [(row, col, piece byte)], hands: [(color, piece type, count)] -> (board, hands, side)"""
    b, h = S.empty_board()
    for r, c, p in pieces:
        b[sq(r, c)] = p
    for color, t, n in hands:
        h[color, HAND[t]] = n
    return b, h, side


KINGS = [(8, 4, KING), (0, 4, KING | WHITE)]

# ---------------------------------------------------------------- rules.rs: piece values (:910-950)
PIECE_VALUES = [  # (type, unpromoted, promoted)   rules.rs:913-933
    (PAWN, 1, 7), (LANCE, 3, 6), (KNIGHT, 4, 6), (SILVER, 5, 6), (GOLD, 6, 6), (BISHOP, 8, 10), (ROOK, 10, 12), (KING, 0, 0)]


def test_piece_value_all_combinations():
    for t, plain, promoted in PIECE_VALUES:
        assert S.piece_value(t, False) == plain and S.piece_value(t, True) == promoted, t
    for t in (PAWN, LANCE, KNIGHT, SILVER, BISHOP, ROOK):                                # rules.rs:940-950
        assert S.piece_value(t, True) > S.piece_value(t, False)


# ---------------------------------------------------------------- positions with their expected numbers
# name -> (board pieces, hands, side, expectations).  Expectation keys:
#   material: material_balance(pos, Black) (rules.rs:356-383);  impasse: compute_impasse_score per colour;
#   in_check: is the side to move in check;  legal: number of legal moves;  impasse_result: check_impasse (result, winner)
POSITIONS = {
    # rules.rs:968-988 / :990-1005  Black has an extra rook: +10, and the balance negates with the perspective
    "extra_rook": (KINGS + [(4, 0, ROOK)], (), 0, dict(material=10)),
    # rules.rs:1007-1022  a gold in hand counts with its board value
    "gold_in_hand": (KINGS, [(0, GOLD, 1)], 0, dict(material=6)),
    # rules.rs:1024-1040  a dragon counts 12, not 10
    "dragon": (KINGS + [(4, 0, ROOK | PROM)], (), 0, dict(material=12)),
    # rules.rs:1042-1055  kings are excluded
    "kings_only": (KINGS, (), 0, dict(material=0)),
    # rules.rs:1072-1086  impasse: a promoted rook is still worth 5
    "impasse_dragon": ([(4, 4, ROOK | PROM), (8, 4, KING)], (), 0, dict(impasse=(5, 0))),
    # rules.rs:1588-1601  a tokin is worth 1, not 5
    "impasse_tokin": ([(4, 4, PAWN | PROM), (8, 4, KING)], (), 0, dict(impasse=(1, 0))),
    # rules.rs:1603-1616  a horse is worth 5
    "impasse_horse": ([(4, 4, BISHOP | PROM), (8, 4, KING)], (), 0, dict(impasse=(5, 0))),
    # rules.rs:1618-1646  rook 5 + horse 5 + tokin 1 + gold 1 + silver in hand 1 = 13
    "impasse_mixed": ([(8, 4, KING), (4, 0, ROOK), (4, 1, BISHOP | PROM), (4, 2, PAWN | PROM), (4, 3, GOLD)], [(0, SILVER, 1)], 0,
                      dict(impasse=(13, 0))),
    # rules.rs:1925-1947  only Black's king has entered (ten black pieces in the zone, 24+ points): no impasse
    "impasse_one_king": ([(0, 4, KING), (0, 0, KING | WHITE), (2, 0, PAWN)] + [(1, c, PAWN) for c in (0, 1, 2, 3, 5, 6, 7, 8)],
                         [(0, ROOK, 3)], 0, dict(impasse_result=(S.R_PROGRESS, -1))),
    # game.rs:2051-2110  White has no legal move and is not in check: the side without moves loses
    "stalemate_white": ([(8, 8, KING | WHITE), (6, 7, KING), (8, 7, PAWN), (7, 8, PAWN), (7, 7, PAWN), (8, 3, ROOK)], (), 1,
                        dict(in_check=False, legal=0)),
    # katago_observation.rs:486-517  Black rook on the king's file, White to move: White is in check
    "white_in_check": ([(0, 4, KING | WHITE), (8, 4, KING), (4, 4, ROOK)], (), 1, dict(in_check=True)),
}


@pytest.mark.parametrize("name", sorted(POSITIONS))
def test_position_known_answers(name):
    pieces, hands, side, want = POSITIONS[name]
    e = _env(*position(pieces, hands, side))
    if "material" in want:
        assert e.material(0, 0) == want["material"] and e.material(0, 1) == -want["material"]      # rules.rs:968-988 antisymmetry
    if "impasse" in want:
        assert (e.impasse_score(0, 0), e.impasse_score(0, 1)) == want["impasse"]
    if "impasse_result" in want:
        assert e.impasse() == want["impasse_result"]
    if "in_check" in want:
        assert e.in_check(0, side) == want["in_check"]
    if "legal" in want:
        assert e.legal_count(0) == want["legal"]


def test_material_and_impasse_at_the_start_position():
    e = OracleVecEnv(1)
    assert e.material(0, 0) == 0 and e.material(0, 1) == 0                               # rules.rs:957-965
    b, h, side, _ = e.state(0)
    h = h.copy(); h[0, HAND[PAWN]] = 2                                                   # rules.rs:1061-1069: 27 + two pawns in hand
    assert _env(b, h, side).impasse_score(0, 0) == 29


# ---------------------------------------------------------------- rules.rs:1792-1919 piece_attacks_square, one table
# (piece byte, from, [(target, attacked?)], blockers [(row, col, piece)])
ATTACKS = [
    (KNIGHT, (4, 4), [((2, 3), True), ((2, 5), True), ((3, 4), False)], []),                                     # :1795-1807
    (KNIGHT | WHITE, (4, 4), [((6, 3), True), ((6, 5), True)], []),                                              # :1809-1819
    (LANCE, (6, 4), [((3, 4), True)], []),                                                                        # :1821-1828
    (LANCE, (6, 4), [((3, 4), False), ((4, 4), True)], [(4, 4, PAWN | WHITE)]),                                  # :1829-1838 blocked
    (SILVER, (4, 4), [((3, 4), True), ((3, 3), True), ((3, 5), True), ((5, 3), True), ((5, 5), True), ((4, 3), False), ((5, 4), False)], []),  # :1840-1854
    (GOLD, (4, 4), [((3, 4), True), ((3, 3), True), ((4, 3), True), ((5, 4), True), ((5, 3), False), ((5, 5), False)], []),                    # :1856-1870
    (BISHOP, (4, 4), [((2, 2), True), ((6, 6), True), ((4, 6), False)], []),                                     # :1872-1882
    (ROOK | PROM, (4, 4), [((4, 8), True), ((0, 4), True), ((3, 3), True), ((5, 5), True), ((2, 2), False)], []),  # :1884-1899
    (BISHOP | PROM, (4, 4), [((2, 2), True), ((3, 4), True), ((4, 5), True), ((2, 4), False)], []),              # :1901-1914
]


@pytest.mark.parametrize("case", range(len(ATTACKS)))
def test_piece_attacks_square(case):
    piece, frm, targets, blockers = ATTACKS[case]
    e = _env(*position(blockers))
    for (r, c), want in targets:
        assert e.piece_attacks(0, sq(*frm), piece, sq(r, c)) == want, (piece, frm, (r, c))


# ---------------------------------------------------------------- game.rs:1915-2050 drops that give check
def test_drops_that_give_check_and_the_pawn_column_rule():
    for piece in (ROOK, GOLD):                                                           # game.rs:1915-1963 / :1965-2006
        e = _env(*position(KINGS, [(0, piece, 2 if piece == GOLD else 1)], 0))
        e.play(0, 0, sq(1, 4), drop=piece)
        _, hands, side, _ = e.state(0)
        assert side == 1 and e.in_check(0, 1)                                            # "White king should be in check"
        assert hands[0, HAND[piece]] == (1 if piece == GOLD else 0)
    # game.rs:2008-2048: no black pawn stands in column 0 before the drop at (5,0); after it the column is taken -- the next
    # pawn drop there is refused (nifu, movegen.rs:166-175): expressed through the legal mask, the engine's column cache is internal
    e = _env(*position(KINGS, [(0, PAWN, 2)], 0))
    drops_col0 = lambda env: [m for m in _moves(env) if m[3] == PAWN and m[1] % 9 == 0]
    assert len(drops_col0(e)) == 8                                                       # rows 1..8 (row 0: no further move)
    e.play(0, 0, sq(5, 0), drop=PAWN)
    e.play(0, sq(0, 4), sq(0, 3))                                                        # White moves; Black again
    assert drops_col0(e) == []


def _moves(e, i=0):
    _, mask = e.observe(i)
    white = bool(e.state(i)[2])
    return [S.decode(int(a), white=white) for a in np.flatnonzero(mask)]


def test_stalemate_is_a_loss_for_the_side_without_moves():
    """game.rs:2051-2110 (and :2112-2158: asking again gives the same answer): reached through a step -- Black's rook arrives
    on (8,3), White has no move and is not in check: Checkmate{winner: Black}, reward +1 for the mover."""
    pieces, hands, _, _ = POSITIONS["stalemate_white"]
    pieces = [p for p in pieces if p[2] != ROOK] + [(0, 3, ROOK)]                        # the rook one move before
    e = _env(*position(pieces, hands, 0))
    out = e.step([S.encode(sq(0, 3), sq(8, 3))])
    assert bool(out["terminated"][0]) and int(out["termination_reason"][0]) == S.R_CHECKMATE and float(out["rewards"][0]) == 1.0
    e2 = _env(*position(POSITIONS["stalemate_white"][0], (), 1))
    assert e2.check_termination() == e2.check_termination() == (S.R_CHECKMATE, 0)


# ---------------------------------------------------------------- vec_env.rs test module
def test_vec_env_rs_masks_and_counters():
    e = OracleVecEnv(1)
    obs, mask = e.reset()
    assert mask[0].sum() == 30 == e.legal_count(0)                                       # vec_env.rs:1083-1101
    for m in _moves(e):                                                                  # :1156-1186 every legal move's index is set
        assert mask[0][S.encode(m[0], m[1], bool(m[2]), m[3])]
    first = int(np.flatnonzero(mask[0])[0])
    out = e.step([first])
    assert out["legal_masks"][0].sum() == 30                                             # :1120-1153 "White's first move should also have 30 options"
    assert out["current_players"].tolist() == [1]                                        # :1400-1430
    assert out["captured_piece"].tolist() == [255] and out["termination_reason"].tolist() == [0]   # :1105-1117 defaults when nothing happens
    assert out["ply_count"].tolist() == [1]
    # :1216-1258 ten plies of first-legal-move play: reward 0 while in progress, mask count == legal count every ply
    e = OracleVecEnv(1)
    _, mask = e.reset()
    for _ in range(10):
        out = e.step([int(np.flatnonzero(mask[0])[0])])
        assert float(out["rewards"][0]) == 0.0 and not out["terminated"][0]
        mask = out["legal_masks"]
        assert mask[0].sum() == e.legal_count(0)
    # :1748-1783 material balance is reported from the mover's point of view: Black captures a pawn -> +1 after Black's move
    b, h = S.empty_board()
    b[sq(8, 4)] = KING; b[sq(0, 4)] = KING | WHITE; b[sq(4, 4)] = ROOK; b[sq(4, 7)] = PAWN | WHITE
    e = _env(b, h, 0)
    out = e.step([S.encode(sq(4, 4), sq(4, 7))])
    assert out["captured_piece"].tolist() == [0] and out["material_balance"].tolist() == [10 + 1]   # rook on the board + pawn in hand


def test_vec_env_rs_draw_rate_and_stats():
    """vec_env.rs:1629-1670, 1380-1398; test_vec_env.py:149-197: rates are 0 before any episode; max_ply = 1 truncates every game
    at once: episodes 2, truncation rate 1, mean length 1; five more one-ply games accumulate; reset_stats clears."""
    e = OracleVecEnv(2, 1, "default", "default")
    st = e.stats()
    assert st["episodes_completed"] == 0 and st["episodes_drawn"] == 0
    _, mask = e.reset()
    out = e.step([int(np.flatnonzero(m)[0]) for m in mask])
    st = e.stats()
    assert st["episodes_completed"] == 2 and st["episodes_truncated"] == 2 and st["total_episode_ply"] == 2
    assert out["truncated"].tolist() == [True, True] and out["terminated"].tolist() == [False, False]
    assert out["rewards"].tolist() == [0.0, 0.0]                                         # vec_env.rs:1047-1052 max moves: 0
    e1 = OracleVecEnv(1, 1, "default", "default")
    _, mask = e1.reset()
    for _ in range(5):
        mask = e1.step([int(np.flatnonzero(mask[0])[0])])["legal_masks"]
    st = e1.stats()
    assert st["episodes_completed"] == 5 and st["total_episode_ply"] == 5               # test_vec_env.py:176-184
    S.lib().so_reset_stats(e1.h)
    assert e1.stats()["episodes_completed"] == 0


def test_observation_planes_at_the_start_position():
    """shogi-gym/tests/test_observation.py:7-37 and test_vec_env.py:336-346 (default 46-plane mode); vec_env.rs:1189-1213."""
    e = OracleVecEnv(1, 100, "default", "default")
    obs, mask = e.reset()
    o = obs[0]
    assert o.dtype == np.float32 and o.shape == (46, 9, 9) and mask.shape == (1, 13527)
    assert o[0:8].sum() > 0 and o[14:22].sum() > 0                                       # own / opponent pieces present
    assert np.all(o[42] == 1.0)                                                          # player indicator: Black
    assert np.all(o[44] == 0.0) and np.all(o[45] == 0.0)                                 # reserved
    assert np.all(o[28:42] == 0.0)                                                       # empty hands
    assert set(np.unique(o[:28]).tolist()) <= {0.0, 1.0}                                 # piece planes are binary
    k = OracleVecEnv(1, 100)                                                             # katago mode: vec_env.rs:1672-1733
    ko, km = k.reset()
    assert ko.shape == (1, 50, 9, 9) and km.shape == (1, 11259) and km[0].sum() == 30
    assert np.array_equal(ko[0, :44], o[:44])                                            # katago_observation.rs:135-172 first 44 planes = default's
    assert np.all(ko[0, 44:48] == 0) and np.all(ko[0, 48] == 0) and np.all(ko[0, 49] == 0)   # :214-231, :327-338, :186-195


def test_check_plane_follows_the_side_to_move():
    """katago_observation.rs:340-408, 527-571 as far as the VecEnv boundary reaches (it always observes for the side to move;
    :486-525 observes for the OTHER side, which no VecEnv call does): plane 48 is all ones exactly when the mover is in check."""
    pieces, hands, side, _ = POSITIONS["white_in_check"]
    obs, _ = _env(*position(pieces, hands, side)).observe(0)
    assert np.all(obs[48] == 1.0)
    obs, _ = _env(*position(pieces, hands, 0)).observe(0)                                # Black to move: Black is not in check
    assert np.all(obs[48] == 0.0)
    obs, _ = _env(*position([(8, 4, KING), (0, 4, KING | WHITE), (4, 4, ROOK | WHITE)], (), 0)).observe(0)   # :527-571
    assert np.all(obs[48] == 1.0)
    # :573-660 two kings only: at most 4 non-zero piece-plane values, empty hand / repetition / check / reserved planes
    obs, _ = _env(*position(KINGS)).observe(0)
    assert np.count_nonzero(obs[:28]) <= 4 and not obs[28:42].any() and not obs[44:50].any()


# ---------------------------------------------------------------- action mappers: both implementations against the same tables
def _mappers():
    from keisei_amd.shogi_gym import DefaultActionMapper, SpatialActionMapper
    sp, df = SpatialActionMapper(), DefaultActionMapper()

    def host(m):
        def enc(frm, to, promote=False, drop=0, white=False):
            return m.encode_drop_move(to, drop - 1, white) if drop else m.encode_board_move(frm, to, promote, white)

        def dec(idx, white=False):
            try:
                d = m.decode(idx, white)
            except ValueError:
                return None
            return (d["to_sq"], d["to_sq"], 0, d["piece_type_idx"] + 1) if d["type"] == "drop" else (d["from_sq"], d["to_sq"], int(d["promote"]), 0)
        return enc, dec

    def oracle(spatial):
        return (lambda frm, to, promote=False, drop=0, white=False: S.encode(frm, to, promote, drop, white, spatial),
                lambda idx, white=False: S.decode(idx, white, spatial))
    return {"spatial/host": (host(sp), 11259), "spatial/oracle": (oracle(True), 11259),
            "default/host": (host(df), 13527), "default/oracle": (oracle(False), 13527)}


MAPPER_CASES = {   # board moves (from, to, promote) the reference round-trips explicitly
    "default": [(0, 1, False), (0, 1, True), (40, 41, False), (40, 39, True), (80, 79, False), (10, 70, False), (70, 10, True)],   # action_mapper.rs:253-262
    "corners": [(f, t, p) for f in (0, 8, 72, 80) for t in (0, 8, 72, 80) if f != t for p in (False, True)],                        # action_mapper.rs:519-540
}


@pytest.mark.parametrize("which", ["spatial/host", "spatial/oracle", "default/host", "default/oracle"])
def test_action_mapper_reference_cases(which):
    (enc, dec), size = _mappers()[which]
    spatial = which.startswith("spatial")
    assert dec(size) is None and dec(size + 1) is None                                   # decode out of range: both mappers' :416-422 / :518-523
    seen = set()
    for white in (False, True):                                                          # drops: all 81 x 7, no collisions, round trip
        for to in range(81):
            for h in range(7):
                idx = enc(0, to, drop=h + 1, white=white)
                if spatial:
                    assert 132 <= idx % 139 <= 138                                       # spatial_action_mapper.rs:396-412 slot range
                else:
                    assert 12960 <= idx < 13527                                          # action_mapper.rs:280-300, test_action_mapper.py:34-38
                m = dec(idx, white)
                assert m[1] == to and m[3] == h + 1
                seen.add((white, idx))
    assert len(seen) == 2 * 567                                                          # spatial_action_mapper.rs:525-537, :600-612
    assert enc(0, 10, drop=ROOK, white=False) != enc(0, 10, drop=ROOK, white=True)       # perspective flip of a drop (:500-516 / :394-414)
    if spatial:
        # spatial_action_mapper.rs:381-394: N, distance 4 from square 40 -> slot 3
        assert enc(40, sq(0, 4)) == 40 * 139 + 3
        # :428-456 every ray square from (4,4), both promotion flags; :458-479 / :567-585 knight jumps incl. the edge columns
        for d, (dr, dc) in enumerate(((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1))):
            for dist in range(1, 9):
                r, c = 4 + dr * dist, 4 + dc * dist
                if 0 <= r < 9 and 0 <= c < 9:
                    for p in (False, True):
                        idx = enc(40, sq(r, c), p)
                        assert idx % 139 == (64 if p else 0) + d * 8 + dist - 1 and dec(idx) == (40, sq(r, c), int(p), 0)
        for frm, to in (((4, 4), (2, 3)), ((4, 4), (2, 5)), ((4, 0), (2, 1)), ((4, 8), (2, 7))):
            for p in (False, True):
                idx = enc(sq(*frm), sq(*to), p)
                assert 128 <= idx % 139 <= 131 and dec(idx) == (sq(*frm), sq(*to), int(p), 0)
        for to in ((8, 5), (8, 3)):                                                      # :614-640 White's knight jumps "down"
            for p in (False, True):
                assert dec(enc(sq(6, 4), sq(*to), p, white=True), True) == (sq(6, 4), sq(*to), int(p), 0)
        assert dec(0 * 139 + 128) is None and dec(0 * 139 + 130) is None                # :587-598 knight from row 0 leaves the board
        for corner in (0, 8, 72, 80):                                                    # :539-565 distance-1 moves from the corners
            for dr, dc in ((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)):
                r, c = corner // 9 + dr, corner % 9 + dc
                if 0 <= r < 9 and 0 <= c < 9:
                    assert dec(enc(corner, sq(r, c))) == (corner, sq(r, c), 0, 0)
        a, b = enc(20, 11), enc(20, 11, white=True)                                      # :481-498
        assert a != b and dec(a) == (20, 11, 0, 0) and dec(b, True) == (20, 11, 0, 0)
    else:
        for frm, to, p in MAPPER_CASES["default"] + MAPPER_CASES["corners"]:
            idx = enc(frm, to, p)
            assert idx < 12960 and dec(idx) == (frm, to, int(p), 0)
        assert enc(0, 1) == enc(80, 79, white=True)                                      # test_action_mapper.py:40-43
        a, b = enc(20, 30), enc(20, 30, white=True)                                      # action_mapper.rs:361-392
        assert a != b and dec(a) == (20, 30, 0, 0) and dec(b, True) == (20, 30, 0, 0)
        assert dec(12959)[3] == 0 and dec(12960)[3] != 0 and dec(13526) is not None      # :489-517 board / drop boundary
        assert all(dec(i) is not None for i in range(0, 13527))                          # :542-551 every index decodes


def test_host_mapper_argument_errors():
    """shogi-gym/tests/test_action_mapper.py:45-56 (PyO3 raises ValueError)."""
    from keisei_amd.shogi_gym import DefaultActionMapper, SpatialActionMapper
    for m in (DefaultActionMapper(), SpatialActionMapper()):
        with pytest.raises(ValueError):
            m.decode(m.action_space_size, False)
        with pytest.raises(ValueError):
            m.encode_board_move(81, 0, False, False)
        with pytest.raises(ValueError):
            m.encode_drop_move(0, 7, False)


def test_legal_moves_round_trip_through_the_spatial_mapper_for_twelve_plies():
    """spatial_action_mapper.rs:660-720: every legal move of twelve plies of first-legal-move play encodes inside the action
    space and decodes back to itself for the side to move (the mask IS the set of encodings, so: decode, re-encode)."""
    e = OracleVecEnv(1)
    _, mask = e.reset()
    for ply in range(12):
        white = bool(e.state(0)[2])
        idxs = np.flatnonzero(mask[0])
        assert len(idxs) > 0
        for a in idxs:
            m = S.decode(int(a), white)
            assert m is not None and S.encode(m[0], m[1], bool(m[2]), m[3], white) == int(a)
        mask = e.step([int(idxs[0])])["legal_masks"]


# ---------------------------------------------------------------- step_result.rs: TerminationReason codes
def test_termination_reason_codes():
    """step_result.rs:103-107 in progress 0, :109-115 checkmate 1, :117-121 repetition 2, :123-129 perpetual check 3,
    :131-137 impasse with a winner 4, :139-143 impasse draw 4, :145-149 max moves 5 -- the u8 the env writes into
    step_metadata.termination_reason (the oracle's result codes are these numbers)."""
    assert (S.R_PROGRESS, S.R_CHECKMATE, S.R_REPETITION, S.R_PERPETUAL, S.R_IMPASSE, S.R_MAXMOVES) == (0, 1, 2, 3, 4, 5)
    e = OracleVecEnv(2, 1)                                           # max_ply 1: the first move truncates -> code 5
    _, mask = e.reset()
    r = e.step(np.array([int(np.flatnonzero(m)[0]) for m in mask], dtype=np.int64))
    assert r["termination_reason"].tolist() == [5, 5] and r["truncated"].tolist() == [True, True]
    e = OracleVecEnv(1, 100)
    _, mask = e.reset()
    r = e.step(np.array([int(np.flatnonzero(mask[0])[0])], dtype=np.int64))
    assert r["termination_reason"].tolist() == [0]


def test_plane_counts_and_action_space_sizes():
    """katago_observation.rs:116-123 (50 channels), :125-132 (buffer 50 * 81 = 4050); observation.rs:248-252 (46 channels);
    action_mapper.rs:241-246 and shogi-gym/tests/test_action_mapper.py:10-11 (13 527); spatial_action_mapper.rs:373-379 (11 259)."""
    for omode, planes in (("katago", 50), ("default", 46)):
        for amode, size in (("spatial", 11259), ("default", 13527)):
            obs, mask = OracleVecEnv(1, 100, omode, amode).reset()
            assert obs.shape == (1, planes, 9, 9) and obs[0].size == planes * 81 and mask.shape == (1, size)
    assert S.A_SIZE == 11259 and S.A_DEFAULT == 13527


def test_white_drops_exclude_dead_squares():
    """movegen.rs:1163-1213: White's pawn / lance never drop on row 8, White's knight never on rows 7-8."""
    e = _env(*position(KINGS, [(1, PAWN, 1), (1, KNIGHT, 1), (1, LANCE, 1)], side=1))
    drops = [(to, kind) for _, to, _, kind in e.pseudo_moves(0, 1, boards_only=False) if kind]       # (from, to, promote, drop type)
    assert {k for _, k in drops} == {PAWN, KNIGHT, LANCE}
    for to, kind in drops:
        row = to // 9
        if kind in (PAWN, LANCE):
            assert row != 8
        if kind == KNIGHT:
            assert row < 7


def test_random_games_stay_consistent():
    """game.rs:2159-2215 plays 100 xorshift-random games of up to 200 plies and recomputes its incremental hash, attack map and
    pawn columns after every move.  The oracle keeps no incremental state (it recomputes everything per call), so what remains
    of that test here is its walk: the same 100 seeds and move choices, every position reached must keep 40 pieces with two kings,
    a legal-move count equal to the mask's popcount, and an attack map that agrees with in_check."""
    for seed in range(100):
        e = OracleVecEnv(1, 200)
        _, mask = e.reset()
        x = seed
        for _ply in range(200):
            legal = np.flatnonzero(mask[0])
            assert len(legal) == e.legal_count(0) > 0
            x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF; x ^= x >> 7; x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
            r = e.step(np.array([int(legal[x % len(legal)])], dtype=np.int64))
            mask = r["legal_masks"]
            if r["terminated"][0] or r["truncated"][0]:
                break
            board, hands, side, _ = e.state(0)
            assert int((board != 0).sum() + hands.sum()) == 40 and int(((board & 15) == KING).sum()) == 2
            amap = e.attack_map(0).reshape(2, 81)
            king = int(np.flatnonzero(board == (KING | (WHITE if side else 0)))[0])
            assert (amap[1 - side][king] > 0) == e.in_check(0, side)
