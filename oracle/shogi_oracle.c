/* TEST INFRASTRUCTURE -- the CPU oracle of SURVEY §8 row f3 (the VecEnv observation / legal-mask producer).
 *
 * A plain-C restatement of the reference's Rust rules engine and vectorised environment, written from a reading of
 *   shogi-engine/crates/shogi-core/src/{types,piece,position,attack,movegen,rules,game}.rs
 *   shogi-engine/crates/shogi-gym/src/{observation,katago_observation,spatial_action_mapper,vec_env,step_result}.rs
 * Every function names the lines it follows.  It deliberately keeps the reference's METHOD (mailbox board, pseudo-legal
 * generation, make the move, recompute the whole attack map, look at the king) so that it is an independent check of the
 * HIP kernels, which decide legality by looking outward from the king instead.  Repetition is decided on whole positions
 * (board, hands, side to move), where the reference compares 64-bit Zobrist keys and the HIP path a 64-bit mixed key.
 *
 * Parity pin: no Rust toolchain exists in the build image, so the reference cannot be run.  The oracle is pinned by the
 * reference's own known answers (tests/test_shogi_oracle.py): perft 30 / 900 / 25 470 / 719 731 from the start position
 * (game.rs:1225-1243, :1900), the mate, stalemate, nifu, pin, check-escape and capture-to-hand positions of game.rs:623-1544,
 * the generator counts of movegen.rs:242-1146 (lone sliders, corners, promotion zones, dead drops), the pawn-drop-mate,
 * repetition, perpetual-check and impasse positions of rules.rs:575-1790, the plane contents of observation.rs:255-724 and
 * katago_observation.rs:214-420, the start-position mask facts and the reward table of vec_env.rs:986-1215, the action-index
 * examples of spatial_action_mapper.rs:357-726 and action_mapper.rs:17-110, and the VecEnv behaviour its Python tests
 * state (shogi-gym/tests/test_vec_env.py: shapes, truncation, restart, counters).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { PAWN = 1, LANCE, KNIGHT, SILVER, GOLD, BISHOP, ROOK, KING };          /* types.rs:56-66 */
#define WHITE_BIT 0x10                                                       /* piece.rs:10-19 */
#define PROM_BIT 0x20
#define TYPE(p) ((p) & 0x0F)
#define COLOR(p) (((p) >> 4) & 1)
#define PROMOTED(p) (((p) & PROM_BIT) != 0)

enum { R_PROGRESS = 0, R_CHECKMATE, R_REPETITION, R_PERPETUAL, R_IMPASSE, R_MAXMOVES };   /* step_result.rs:9-16 */

#define A_TYPES 139
#define A_SPATIAL (81 * A_TYPES)
#define A_BOARD (81 * 80 * 2)                 /* action_mapper.rs:17-19 */
#define A_DEFAULT (A_BOARD + 81 * 7)
#define A_MAX A_DEFAULT
static int a_size(int amode) { return amode ? A_SPATIAL : A_DEFAULT; }
static int obs_len(int omode) { return (omode ? 50 : 46) * 81; }

typedef struct { uint8_t board[81]; uint8_t hands[2][7]; uint8_t side; } Pos;        /* position.rs:20-25 */
typedef struct { uint8_t from, to, promote, drop; } Mv;                              /* drop: 0 = board move, 1..7 hand type */

typedef struct {
    Pos pos;
    uint32_t ply, max_ply;
    int result, winner;              /* winner: -1 none / draw, 0 black, 1 white */
    Pos* hist;                       /* hist[k] = the position BEFORE move k (game.rs hash_history) */
    uint8_t* in_check_hist;          /* game.rs check_history */
} Game;

typedef struct {
    int n; uint32_t max_ply; Game* g;
    int omode, amode;                /* 1 = katago 50 planes / spatial 11 259 actions; 0 = the 46-plane / 13 527-action defaults */
    uint8_t* mask;                   /* the masks handed out last (step() validates against them, vec_env.rs:666-672) */
    uint64_t stats[4];               /* completed, drawn, truncated, total ply (vec_env.rs:395-408) */
} Env;

static int pos_equal(const Pos* a, const Pos* b) {
    return memcmp(a->board, b->board, 81) == 0 && memcmp(a->hands, b->hands, 14) == 0 && a->side == b->side;
}

/* ------------------------------------------------------------------------------------------------ attack.rs */
static const int8_t kDR[8] = {-1, -1, 0, 1, 1, 1, 0, -1};      /* N NE E SE S SW W NW (spatial_action_mapper.rs:31-40) */
static const int8_t kDC[8] = {0, 1, 1, 1, 0, -1, -1, -1};

/* the step and slide directions of a piece, as bit sets over the eight directions above, for BLACK (attack.rs:56-113);
 * white is the same set turned by 180 degrees */
static void piece_dirs(int type, int promoted, int color, unsigned* steps, unsigned* slides) {
    const unsigned N = 1, NE = 2, E = 4, SE = 8, S = 16, SW = 32, W = 64, NW = 128;
    const unsigned gold = N | NE | NW | E | W | S;
    unsigned st = 0, sl = 0;
    if (promoted) {
        if (type == PAWN || type == LANCE || type == KNIGHT || type == SILVER) st = gold;
        else if (type == BISHOP) { st = N | E | S | W; sl = NE | SE | SW | NW; }
        else if (type == ROOK) { st = NE | SE | SW | NW; sl = N | E | S | W; }
    } else switch (type) {
        case PAWN: st = N; break;
        case LANCE: sl = N; break;
        case KNIGHT: break;
        case SILVER: st = N | NE | NW | SE | SW; break;
        case GOLD: st = gold; break;
        case BISHOP: sl = NE | SE | SW | NW; break;
        case ROOK: sl = N | E | S | W; break;
        case KING: st = 255; break;
    }
    if (color) { st = ((st << 4) | (st >> 4)) & 255; sl = ((sl << 4) | (sl >> 4)) & 255; }
    *steps = st; *slides = sl;
}

static int knight_targets(int sq, int color, int out[2]) {          /* attack.rs:119-139 */
    const int row = sq / 9, col = sq % 9, tr = color ? row + 2 : row - 2;
    int n = 0;
    for (int dc = -1; dc <= 1; dc += 2) {
        const int tc = col + dc;
        if (tr >= 0 && tr < 9 && tc >= 0 && tc < 9) out[n++] = tr * 9 + tc;
    }
    return n;
}

static void add_sat(uint8_t* v) { if (*v < 255) ++*v; }

static void attack_map(const Pos* p, uint8_t map[2][81]) {          /* attack.rs:145-200 */
    memset(map, 0, 2 * 81);
    for (int sq = 0; sq < 81; ++sq) {
        const int pc = p->board[sq];
        if (!pc) continue;
        const int c = COLOR(pc), t = TYPE(pc), pr = PROMOTED(pc);
        if (t == KNIGHT && !pr) {
            int tg[2]; const int n = knight_targets(sq, c, tg);
            for (int i = 0; i < n; ++i) add_sat(&map[c][tg[i]]);
            continue;
        }
        unsigned st, sl; piece_dirs(t, pr, c, &st, &sl);
        for (int d = 0; d < 8; ++d) {
            if (!(((st | sl) >> d) & 1)) continue;
            int r = sq / 9 + kDR[d], cc = sq % 9 + kDC[d];
            while (r >= 0 && r < 9 && cc >= 0 && cc < 9) {
                add_sat(&map[c][r * 9 + cc]);
                if (!((sl >> d) & 1) || p->board[r * 9 + cc]) break;
                r += kDR[d]; cc += kDC[d];
            }
        }
    }
}

static int find_king(const Pos* p, int color) {                     /* position.rs:141-150 */
    const int target = KING | (color ? WHITE_BIT : 0);
    for (int i = 0; i < 81; ++i) if (p->board[i] == target) return i;
    return -1;
}

static int color_in_check(const Pos* p, int color) {                /* game.rs:98-105 */
    const int k = find_king(p, color);
    if (k < 0) return 0;
    uint8_t map[2][81]; attack_map(p, map);
    return map[color ^ 1][k] > 0;
}

/* ------------------------------------------------------------------------------------------------ movegen.rs */
static int in_zone(int row, int color) { return color ? row >= 6 : row <= 2; }                      /* :19-24 */
static int must_promote(int type, int to_row, int color) {                                          /* :33-45 */
    if (type == PAWN || type == LANCE) return color ? to_row == 8 : to_row == 0;
    if (type == KNIGHT) return color ? to_row >= 7 : to_row <= 1;
    return 0;
}

static int push_board_move(Mv* out, int n, int from, int to, int type, int promoted, int color) {   /* :77-103 */
    const int can = type != GOLD && type != KING;
    if (promoted || !can) { out[n++] = (Mv){(uint8_t)from, (uint8_t)to, 0, 0}; return n; }
    if (must_promote(type, to / 9, color)) out[n++] = (Mv){(uint8_t)from, (uint8_t)to, 1, 0};
    else if (in_zone(from / 9, color) || in_zone(to / 9, color)) {
        out[n++] = (Mv){(uint8_t)from, (uint8_t)to, 0, 0};
        out[n++] = (Mv){(uint8_t)from, (uint8_t)to, 1, 0};
    } else out[n++] = (Mv){(uint8_t)from, (uint8_t)to, 0, 0};
    return n;
}

static int pseudo_board_moves(const Pos* p, int color, Mv* out, int n) {                            /* :112-177 */
    for (int from = 0; from < 81; ++from) {
        const int pc = p->board[from];
        if (!pc || COLOR(pc) != color) continue;
        const int t = TYPE(pc), pr = PROMOTED(pc);
        if (t == KNIGHT && !pr) {
            int tg[2]; const int k = knight_targets(from, color, tg);
            for (int i = 0; i < k; ++i) {
                const int q = p->board[tg[i]];
                if (q && COLOR(q) == color) continue;
                n = push_board_move(out, n, from, tg[i], t, pr, color);
            }
            continue;
        }
        unsigned st, sl; piece_dirs(t, pr, color, &st, &sl);
        for (int d = 0; d < 8; ++d) {
            if (!(((st | sl) >> d) & 1)) continue;
            int r = from / 9 + kDR[d], c = from % 9 + kDC[d];
            while (r >= 0 && r < 9 && c >= 0 && c < 9) {
                const int to = r * 9 + c, q = p->board[to];
                if (q && COLOR(q) == color) break;
                n = push_board_move(out, n, from, to, t, pr, color);
                if (q || !((sl >> d) & 1)) break;
                r += kDR[d]; c += kDC[d];
            }
        }
    }
    return n;
}

static int pseudo_drops(const Pos* p, int color, Mv* out, int n) {                                  /* :182-203 */
    for (int h = 0; h < 7; ++h) {
        if (!p->hands[color][h]) continue;
        for (int to = 0; to < 81; ++to) {
            if (p->board[to]) continue;
            if (must_promote(h + 1, to / 9, color)) continue;          /* is_dead_drop: the same table (:50-62) */
            out[n++] = (Mv){0, (uint8_t)to, 0, (uint8_t)(h + 1)};
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------ game.rs */
static int make_move(Game* g, Mv m, int record) {          /* game.rs:107-188; returns the captured piece byte (0 none) */
    Pos* p = &g->pos;
    if (record) {
        g->hist[g->ply] = *p;
        g->in_check_hist[g->ply] = (uint8_t)color_in_check(p, p->side);
    }
    const int me = p->side;
    int captured = 0;
    if (!m.drop) {
        const int pc = p->board[m.from];
        p->board[m.from] = 0;
        captured = p->board[m.to];
        if (captured) p->hands[me][TYPE(captured) - 1]++;
        p->board[m.to] = (uint8_t)(m.promote ? (pc | PROM_BIT) : pc);
    } else {
        p->hands[me][m.drop - 1]--;
        p->board[m.to] = (uint8_t)(m.drop | (me ? WHITE_BIT : 0));
    }
    p->side ^= 1;
    g->ply++;
    return captured;
}

static int piece_attacks_square(const Pos* p, int from, int pc, int target) {                       /* rules.rs:136-176 */
    const int t = TYPE(pc), c = COLOR(pc), pr = PROMOTED(pc);
    if (t == KNIGHT && !pr) {
        int tg[2]; const int n = knight_targets(from, c, tg);
        for (int i = 0; i < n; ++i) if (tg[i] == target) return 1;
        return 0;
    }
    unsigned st, sl; piece_dirs(t, pr, c, &st, &sl);
    for (int d = 0; d < 8; ++d) {
        if (!(((st | sl) >> d) & 1)) continue;
        int r = from / 9 + kDR[d], cc = from % 9 + kDC[d];
        while (r >= 0 && r < 9 && cc >= 0 && cc < 9) {
            if (r * 9 + cc == target) return 1;
            if (!((sl >> d) & 1) || p->board[r * 9 + cc]) break;
            r += kDR[d]; cc += kDC[d];
        }
    }
    return 0;
}

static int is_uchi_fu_zume(const Pos* pos, int to, int color) {                                     /* rules.rs:18-131 */
    Pos p = *pos;
    const int opp = color ^ 1;
    p.board[to] = (uint8_t)(PAWN | (color ? WHITE_BIT : 0));       /* hand and side to move are left alone, as there */
    uint8_t map[2][81]; attack_map(&p, map);
    const int k = find_king(&p, opp);
    if (k < 0 || map[color][k] == 0) return 0;
    for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {       /* 1. the king steps away (or takes) */
        if (!dr && !dc) continue;
        const int r = k / 9 + dr, c = k % 9 + dc;
        if (r < 0 || r > 8 || c < 0 || c > 8) continue;
        const int q = p.board[r * 9 + c];
        if (q && COLOR(q) == opp) continue;
        if (map[color][r * 9 + c] > 0) continue;
        return 0;
    }
    for (int sq = 0; sq < 81; ++sq) {                                          /* 2. another piece takes the pawn */
        const int pc = p.board[sq];
        if (!pc || COLOR(pc) != opp || TYPE(pc) == KING) continue;
        if (!piece_attacks_square(&p, sq, pc, to)) continue;
        Pos s = p;
        s.board[sq] = 0; s.board[to] = (uint8_t)pc;
        uint8_t m2[2][81]; attack_map(&s, m2);
        if (m2[color][k] == 0) return 0;
    }
    return 1;
}

static int legal_moves(const Game* g, Mv* out) {                                                    /* game.rs:262-335 */
    static _Thread_local Mv cand[2048];
    const Pos* p = &g->pos;
    const int me = p->side;
    int nc = pseudo_board_moves(p, me, cand, 0);
    nc = pseudo_drops(p, me, cand, nc);
    int n = 0;
    for (int i = 0; i < nc; ++i) {
        const Mv m = cand[i];
        if (m.drop == PAWN) {
            int nifu = 0;                                               /* compute_pawn_columns, game.rs:24-34 */
            for (int r = 0; r < 9; ++r) if (p->board[r * 9 + m.to % 9] == (PAWN | (me ? WHITE_BIT : 0))) nifu = 1;
            if (nifu) continue;
            if (is_uchi_fu_zume(p, m.to, me)) continue;
        }
        Game t; t.pos = *p; t.ply = 0;
        make_move(&t, m, 0);
        const int k = find_king(&t.pos, me);
        int safe = 0;
        if (k >= 0) { uint8_t map[2][81]; attack_map(&t.pos, map); safe = map[me ^ 1][k] == 0; }
        if (safe) out[n++] = m;
    }
    return n;
}

static int repetition_count(const Game* g) {            /* the reference's repetition_map entry of the current position */
    int n = 1;
    for (uint32_t k = 0; k < g->ply; ++k) n += pos_equal(&g->hist[k], &g->pos);
    return n;
}

static int check_sennichite(const Game* g, int* winner) {                                           /* rules.rs:190-235 */
    if (repetition_count(g) < 4) return R_PROGRESS;
    int all_checks = 1, any = 0;
    for (uint32_t k = 0; k < g->ply; ++k)
        if (pos_equal(&g->hist[k], &g->pos)) { any = 1; if (!g->in_check_hist[k]) all_checks = 0; }
    if (!any) return R_REPETITION;
    if (all_checks) { *winner = g->pos.side; return R_PERPETUAL; }      /* the side that was being checked wins */
    return R_REPETITION;
}

static int impasse_value(int type) { return type == ROOK || type == BISHOP ? 5 : type == KING ? 0 : 1; }   /* rules.rs:318-324 */

static int impasse_score(const Pos* p, int color) {                                                 /* rules.rs:291-315 */
    int s = 0;
    for (int i = 0; i < 81; ++i) { const int pc = p->board[i]; if (pc && COLOR(pc) == color) s += impasse_value(TYPE(pc)); }
    for (int h = 0; h < 7; ++h) s += p->hands[color][h] * impasse_value(h + 1);
    return s;
}

static int zone_count(const Pos* p, int color) {                                                    /* rules.rs:267-285 */
    int n = 0;
    for (int i = 0; i < 81; ++i) { const int pc = p->board[i]; if (pc && COLOR(pc) == color && in_zone(i / 9, color)) ++n; }
    return n;
}

static int check_impasse(const Pos* p, int* winner) {                                               /* rules.rs:228-262 */
    const int bk = find_king(p, 0), wk = find_king(p, 1);
    if (bk < 0 || wk < 0) return R_PROGRESS;
    if (bk / 9 > 2 || wk / 9 < 6) return R_PROGRESS;
    if (zone_count(p, 0) < 10 || zone_count(p, 1) < 10) return R_PROGRESS;
    const int bs = impasse_score(p, 0), ws = impasse_score(p, 1);
    if (bs >= 24 && ws >= 24) { *winner = -1; return R_IMPASSE; }
    if (bs >= 24) { *winner = 0; return R_IMPASSE; }
    if (ws >= 24) { *winner = 1; return R_IMPASSE; }
    return R_PROGRESS;
}

static void check_termination(Game* g) {                                                            /* game.rs:355-387 */
    if (g->result != R_PROGRESS) return;
    if (g->ply >= g->max_ply) { g->result = R_MAXMOVES; g->winner = -1; return; }
    int w = -1;
    int r = check_sennichite(g, &w);
    if (r != R_PROGRESS) { g->result = r; g->winner = w; return; }
    r = check_impasse(&g->pos, &w);
    if (r != R_PROGRESS) { g->result = r; g->winner = w; return; }
    static _Thread_local Mv mv[1024];
    if (legal_moves(g, mv) == 0) { g->result = R_CHECKMATE; g->winner = g->pos.side ^ 1; }   /* no moves at all loses too */
}

static void start_position(Pos* p) {                                                                /* position.rs:45-93 */
    static const uint8_t back[9] = {LANCE, KNIGHT, SILVER, GOLD, KING, GOLD, SILVER, KNIGHT, LANCE};
    memset(p, 0, sizeof *p);
    for (int c = 0; c < 9; ++c) {
        p->board[c] = back[c] | WHITE_BIT; p->board[18 + c] = PAWN | WHITE_BIT;
        p->board[54 + c] = PAWN; p->board[72 + c] = back[c];
    }
    p->board[9 + 1] = ROOK | WHITE_BIT; p->board[9 + 7] = BISHOP | WHITE_BIT;
    p->board[63 + 1] = BISHOP; p->board[63 + 7] = ROOK;
}

static void game_reset(Game* g) { start_position(&g->pos); g->ply = 0; g->result = R_PROGRESS; g->winner = -1; }

/* ------------------------------------------------------------------------------------------------ shogi-gym */
static int piece_value(int type, int promoted) {                                                    /* rules.rs:333-350 */
    static const int plain[9] = {0, 1, 3, 4, 5, 6, 8, 10, 0}, prom[9] = {0, 7, 6, 6, 6, 6, 10, 12, 0};
    return promoted ? prom[type] : plain[type];
}

static int material_balance(const Pos* p, int who) {                                                /* rules.rs:356-383 */
    int b = 0;
    for (int i = 0; i < 81; ++i) {
        const int pc = p->board[i];
        if (!pc || TYPE(pc) == KING) continue;
        const int v = piece_value(TYPE(pc), PROMOTED(pc));
        b += COLOR(pc) == who ? v : -v;
    }
    for (int h = 0; h < 7; ++h) b += piece_value(h + 1, 0) * ((int)p->hands[who][h] - (int)p->hands[who ^ 1][h]);
    return b;
}

static int encode_default(Mv m, int persp) {                                     /* action_mapper.rs:36-45, 63-77 */
    const int t = persp ? 80 - m.to : m.to;
    if (m.drop) return A_BOARD + t * 7 + (m.drop - 1);
    const int f = persp ? 80 - m.from : m.from;
    return f * 160 + (t > f ? t - 1 : t) * 2 + (m.promote ? 1 : 0);
}

static int decode_default(int idx, int persp, Mv* out) {                         /* action_mapper.rs:46-59, 79-110 */
    if (idx < 0 || idx >= A_DEFAULT) return -1;
    if (idx >= A_BOARD) {
        const int t = (idx - A_BOARD) / 7;
        *out = (Mv){0, (uint8_t)(persp ? 80 - t : t), 0, (uint8_t)((idx - A_BOARD) % 7 + 1)};
        return 0;
    }
    const int f = idx / 160, rem = idx % 160, off = rem / 2, t = off >= f ? off + 1 : off;
    *out = (Mv){(uint8_t)(persp ? 80 - f : f), (uint8_t)(persp ? 80 - t : t), (uint8_t)(rem & 1), 0};
    return 0;
}

static int encode_spatial(Mv m, int persp) {                                     /* spatial_action_mapper.rs:138-186 */
    if (m.drop) return (persp ? 80 - m.to : m.to) * A_TYPES + 132 + (m.drop - 1);
    const int f = persp ? 80 - m.from : m.from, t = persp ? 80 - m.to : m.to;
    const int dr = t / 9 - f / 9, dc = t % 9 - f % 9;
    const int adr = abs(dr), adc = abs(dc);
    if ((dr == 0 || dc == 0 || adr == adc) && (adr || adc)) {
        const int ur = (dr > 0) - (dr < 0), uc = (dc > 0) - (dc < 0);
        int dir = 0;
        while (kDR[dir] != ur || kDC[dir] != uc) ++dir;
        const int dist = adr > adc ? adr : adc;
        return f * A_TYPES + (m.promote ? 64 : 0) + dir * 8 + dist - 1;
    }
    if (adr == 2 && adc == 1) {                     /* knight: same sign of dr and dc = slot 0 (":109-133") */
        const int same = (dr > 0 && dc > 0) || (dr < 0 && dc < 0);
        return f * A_TYPES + 128 + (same ? 0 : 1) * 2 + (m.promote ? 1 : 0);
    }
    return -1;
}

static int decode_spatial(int idx, int persp, Mv* out) {                         /* spatial_action_mapper.rs:188-279 */
    if (idx < 0 || idx >= A_SPATIAL) return -1;
    const int sq = idx / A_TYPES, slot = idx % A_TYPES;
    const int row = sq / 9, col = sq % 9;
    if (slot < 132) {
        int tr, tc, promote;
        if (slot < 128) {
            promote = slot >= 64;
            const int b = slot & 63, dir = b / 8, dist = b % 8 + 1;
            tr = row + kDR[dir] * dist; tc = col + kDC[dir] * dist;
        } else {
            const int k = slot - 128;
            promote = k & 1; tr = row - 2; tc = col + (k / 2 == 0 ? -1 : 1);
        }
        if (tr < 0 || tr > 8 || tc < 0 || tc > 8) return -1;
        const int to = tr * 9 + tc;
        *out = (Mv){(uint8_t)(persp ? 80 - sq : sq), (uint8_t)(persp ? 80 - to : to), (uint8_t)promote, 0};
        return 0;
    }
    *out = (Mv){0, (uint8_t)(persp ? 80 - sq : sq), 0, (uint8_t)(slot - 132 + 1)};
    return 0;
}

static int encode_action(Mv m, int persp, int amode) { return amode ? encode_spatial(m, persp) : encode_default(m, persp); }
static int decode_action(int idx, int persp, Mv* out, int amode) { return amode ? decode_spatial(idx, persp, out) : decode_default(idx, persp, out); }

static void write_mask(const Game* g, uint8_t* mask, int amode) {                                              /* vec_env.rs:229-247 */
    static _Thread_local Mv mv[1024];
    memset(mask, 0, (size_t)a_size(amode));
    const int n = legal_moves(g, mv);
    for (int i = 0; i < n; ++i) mask[encode_action(mv[i], g->pos.side, amode)] = 1;
}

static void fill_plane(float* obs, int ch, float v) { for (int i = 0; i < 81; ++i) obs[ch * 81 + i] = v; }

static void write_obs(const Game* g, int persp, float* obs, int omode) {        /* observation.rs:81-153 + katago_observation.rs:41-92 */
    static const int unprom_ch[9] = {0, 0, 1, 2, 3, 4, 5, 6, 7}, prom_ch[9] = {0, 0, 1, 2, 3, 0, 4, 5, 0};
    static const float hand_max[7] = {18.f, 4.f, 4.f, 4.f, 4.f, 2.f, 2.f};
    const Pos* p = &g->pos;
    memset(obs, 0, (size_t)obs_len(omode) * sizeof(float));
    for (int i = 0; i < 81; ++i) {
        const int pc = p->board[i];
        if (!pc) continue;
        const int o = persp ? 80 - i : i, mine = COLOR(pc) == persp;
        const int ch = PROMOTED(pc) ? (mine ? 8 : 22) + prom_ch[TYPE(pc)] : (mine ? 0 : 14) + unprom_ch[TYPE(pc)];
        obs[ch * 81 + o] = 1.f;
    }
    for (int h = 0; h < 7; ++h) {
        fill_plane(obs, 28 + h, (float)p->hands[persp][h] / hand_max[h]);
        fill_plane(obs, 35 + h, (float)p->hands[persp ^ 1][h] / hand_max[h]);
    }
    fill_plane(obs, 42, persp == 0 ? 1.f : 0.f);
    float mc = g->max_ply == 0 ? 0.f : (float)g->ply / (float)g->max_ply;
    if (mc < 0.f) mc = 0.f;
    if (mc > 1.f) mc = 1.f;
    fill_plane(obs, 43, mc);
    if (!omode) return;                                   /* observation.rs: planes 44-45 stay zero */
    const int prior = repetition_count(g) - 1;
    if (prior >= 1 && prior <= 3) fill_plane(obs, 44 + prior - 1, 1.f);
    else if (prior >= 4) fill_plane(obs, 47, 1.f);
    if (color_in_check(p, persp)) fill_plane(obs, 48, 1.f);
}

static float reward_of(int result, int winner, int last_mover) {                                    /* vec_env.rs:98-124 */
    if (result == R_CHECKMATE || result == R_PERPETUAL || (result == R_IMPASSE && winner >= 0)) return winner == last_mover ? 1.f : -1.f;
    return 0.f;
}

/* ------------------------------------------------------------------------------------------------ exported API */
void* so_create(int n, int max_ply, int omode, int amode) {
    Env* e = calloc(1, sizeof *e);
    e->n = n; e->max_ply = (uint32_t)max_ply; e->omode = omode; e->amode = amode;
    e->g = calloc((size_t)n, sizeof(Game));
    e->mask = calloc((size_t)n, A_MAX);
    for (int i = 0; i < n; ++i) {
        e->g[i].max_ply = (uint32_t)max_ply;
        e->g[i].hist = calloc((size_t)max_ply + 2, sizeof(Pos));
        e->g[i].in_check_hist = calloc((size_t)max_ply + 2, 1);
        game_reset(&e->g[i]);
    }
    return e;
}

void so_destroy(void* h) {
    Env* e = h; const size_t OL = (size_t)obs_len(e->omode); const long AS = a_size(e->amode); (void)OL; (void)AS;
    for (int i = 0; i < e->n; ++i) { free(e->g[i].hist); free(e->g[i].in_check_hist); }
    free(e->g); free(e->mask); free(e);
}

void so_reset(void* h, float* obs, uint8_t* mask) {                                                 /* vec_env.rs:617-645 */
    Env* e = h; const size_t OL = (size_t)obs_len(e->omode); const long AS = a_size(e->amode); (void)OL; (void)AS;
    for (int i = 0; i < e->n; ++i) {
        game_reset(&e->g[i]);
        write_obs(&e->g[i], e->g[i].pos.side, obs + (size_t)i * OL, e->omode);
        write_mask(&e->g[i], e->mask + (size_t)i * AS, e->amode);
    }
    memcpy(mask, e->mask, (size_t)e->n * AS);
}

/* vec_env.rs:651-700 (validation: nothing is mutated when any action is refused) and :340-460 (apply_moves).
 * Returns 0, or -(1 + i) for the first env i whose action is negative, undecodable or not in the mask handed out last. */
int so_step(void* h, const int64_t* actions, float* obs, uint8_t* mask, float* rewards, uint8_t* terminated,
            uint8_t* truncated, float* terminal_obs, uint8_t* current_players, uint8_t* captured, uint8_t* term_reason,
            uint16_t* ply, int32_t* material) {
    Env* e = h; const size_t OL = (size_t)obs_len(e->omode); const long AS = a_size(e->amode); (void)OL; (void)AS;
    for (int i = 0; i < e->n; ++i) {
        Mv m;
        if (actions[i] < 0 || decode_action((int)actions[i], e->g[i].pos.side, &m, e->amode) != 0) return -(1 + i);
        if (actions[i] >= AS || !e->mask[(size_t)i * AS + actions[i]]) return -(1 + i);
    }
    /* games are independent (the reference steps them with rayon, vec_env.rs:541-545); OMP_NUM_THREADS sets the cores used */
    int nthreads = e->n / 16;                              /* a thread per 16 games, at most 16 (small test batches stay serial) */
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = 0; i < e->n; ++i) {
        Game* g = &e->g[i];
        Mv m; decode_action((int)actions[i], g->pos.side, &m, e->amode);
        const int cap = make_move(g, m, 1);
        const int last_mover = g->pos.side ^ 1;
        check_termination(g);
        const int term = g->result != R_PROGRESS && g->result != R_MAXMOVES, trunc = g->result == R_MAXMOVES;
        terminated[i] = (uint8_t)term; truncated[i] = (uint8_t)trunc;
        rewards[i] = reward_of(g->result, g->winner, last_mover);
        term_reason[i] = (uint8_t)g->result;
        ply[i] = (uint16_t)g->ply;
        material[i] = material_balance(&g->pos, last_mover);
        captured[i] = cap ? (uint8_t)(TYPE(cap) - 1) : 255;
        if (term || trunc) {
#pragma omp atomic
            e->stats[0]++;
#pragma omp atomic
            e->stats[3] += g->ply;
            if (g->result == R_REPETITION || (g->result == R_IMPASSE && g->winner < 0)) {
#pragma omp atomic
                e->stats[1]++;
            }
            if (g->result == R_MAXMOVES) {
#pragma omp atomic
                e->stats[2]++;
            }
            write_obs(g, g->pos.side, terminal_obs + (size_t)i * OL, e->omode);
            game_reset(g);
        }
        write_obs(g, g->pos.side, obs + (size_t)i * OL, e->omode);
        write_mask(g, e->mask + (size_t)i * AS, e->amode);
        current_players[i] = g->pos.side;
    }
    memcpy(mask, e->mask, (size_t)e->n * AS);
    return 0;
}

void so_stats(void* h, uint64_t out[4]) { memcpy(out, ((Env*)h)->stats, sizeof(uint64_t) * 4); }
void so_reset_stats(void* h) { memset(((Env*)h)->stats, 0, sizeof(uint64_t) * 4); }

void so_get_state(void* h, int i, uint8_t* board, uint8_t* hands, int* side, int* ply) {
    Game* g = &((Env*)h)->g[i];
    memcpy(board, g->pos.board, 81); memcpy(hands, g->pos.hands, 14); *side = g->pos.side; *ply = (int)g->ply;
}

/* place an arbitrary position in env i (the fixtures of rules.rs build theirs square by square); history is cleared */
void so_set_state(void* h, int i, const uint8_t* board, const uint8_t* hands, int side) {
    Env* e = h; Game* g = &e->g[i]; const size_t OL = (size_t)obs_len(e->omode); const long AS = a_size(e->amode); (void)OL; (void)AS;
    memcpy(g->pos.board, board, 81); memcpy(g->pos.hands, hands, 14); g->pos.side = (uint8_t)side;
    g->ply = 0; g->result = R_PROGRESS; g->winner = -1;
    write_mask(g, e->mask + (size_t)i * AS, e->amode);
}

void so_observe(void* h, int i, float* obs, uint8_t* mask) {
    Env* e = h; Game* g = &e->g[i]; const size_t OL = (size_t)obs_len(e->omode); const long AS = a_size(e->amode); (void)OL; (void)AS;
    write_obs(g, g->pos.side, obs, e->omode);
    write_mask(g, e->mask + (size_t)i * AS, e->amode);
    memcpy(mask, e->mask + (size_t)i * AS, (size_t)AS);
}

/* pseudo-legal moves of `color` (movegen.rs:112-203): out[k] = from | to << 8 | promote << 16 | drop << 24; returns the count */
int so_pseudo_moves(void* h, int i, int color, int boards_only, uint32_t* out) {
    static _Thread_local Mv mv[2048];
    const Pos* p = &((Env*)h)->g[i].pos;
    int n = pseudo_board_moves(p, color, mv, 0);
    if (!boards_only) n = pseudo_drops(p, color, mv, n);
    for (int k = 0; k < n; ++k) out[k] = (uint32_t)mv[k].from | ((uint32_t)mv[k].to << 8) | ((uint32_t)mv[k].promote << 16) | ((uint32_t)mv[k].drop << 24);
    return n;
}
/* the 2 x 81 attack counts of the position (attack.rs:145-200) */
void so_attack_map(void* h, int i, uint8_t* out) { uint8_t map[2][81]; attack_map(&((Env*)h)->g[i].pos, map); memcpy(out, map, 162); }
int so_legal_count(void* h, int i) { static _Thread_local Mv mv[1024]; return legal_moves(&((Env*)h)->g[i], mv); }
int so_in_check(void* h, int i, int color) { return color_in_check(&((Env*)h)->g[i].pos, color); }
int so_uchi_fu_zume(void* h, int i, int to, int color) { return is_uchi_fu_zume(&((Env*)h)->g[i].pos, to, color); }
int so_impasse(void* h, int i, int* winner) { *winner = -1; return check_impasse(&((Env*)h)->g[i].pos, winner); }
int so_impasse_score(void* h, int i, int color) { return impasse_score(&((Env*)h)->g[i].pos, color); }
int so_zone_count(void* h, int i, int color) { return zone_count(&((Env*)h)->g[i].pos, color); }
int so_material(void* h, int i, int who) { return material_balance(&((Env*)h)->g[i].pos, who); }
int so_piece_value(int type, int promoted) { return piece_value(type, promoted); }                  /* rules.rs:333-350 */
/* does the piece byte `pc`, standing on `from` of game i's board, attack `target`?  (rules.rs:136-176; tests :1792-1919) */
int so_piece_attacks(void* h, int i, int from, int pc, int target) { return piece_attacks_square(&((Env*)h)->g[i].pos, from, pc, target); }
int so_result(void* h, int i, int* winner) { Game* g = &((Env*)h)->g[i]; *winner = g->winner; return g->result; }
float so_reward(int result, int winner, int last_mover) { return reward_of(result, winner, last_mover); }
int so_encode(int from, int to, int promote, int drop, int persp, int amode) { return encode_action((Mv){(uint8_t)from, (uint8_t)to, (uint8_t)promote, (uint8_t)drop}, persp, amode); }
int so_decode(int idx, int persp, int amode, int out[4]) {
    Mv m; const int rc = decode_action(idx, persp, &m, amode);
    if (rc == 0) { out[0] = m.from; out[1] = m.to; out[2] = m.promote; out[3] = m.drop; }
    return rc;
}

/* the reference's rule tests drive GameState directly (rules.rs:692-905): make_move without the environment
 * bookkeeping, then check_sennichite / check_termination */
void so_play(void* h, int i, int from, int to, int promote, int drop) {
    make_move(&((Env*)h)->g[i], (Mv){(uint8_t)from, (uint8_t)to, (uint8_t)promote, (uint8_t)drop}, 1);
}
int so_sennichite(void* h, int i, int* winner) { *winner = -1; return check_sennichite(&((Env*)h)->g[i], winner); }
int so_check_termination(void* h, int i, int* winner) {
    Game* g = &((Env*)h)->g[i];
    check_termination(g);
    *winner = g->winner;
    return g->result;
}
int so_repetition_count(void* h, int i) { return repetition_count(&((Env*)h)->g[i]); }

static uint64_t perft_rec(Game* g, int depth) {                                                     /* game.rs:1206-1222 */
    Mv mv[1024];
    const int n = legal_moves(g, mv);
    if (depth == 1) return (uint64_t)n;
    uint64_t nodes = 0;
    for (int i = 0; i < n; ++i) {
        const Pos keep = g->pos; const uint32_t ply = g->ply;
        make_move(g, mv[i], 1);
        nodes += perft_rec(g, depth - 1);
        g->pos = keep; g->ply = ply;
    }
    return nodes;
}

uint64_t so_perft(void* h, int i, int depth) { return depth <= 0 ? 1 : perft_rec(&((Env*)h)->g[i], depth); }
