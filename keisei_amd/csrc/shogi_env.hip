// the vectorised shogi environment (observations + legal masks) resident on the device.
//
// The reference steps N games on host cores (shogi-engine/crates/shogi-gym/src/vec_env.rs:340-460, rayon) and hands
// numpy arrays to the trainer, which copies 16 KB of observation and 11 KB of mask per game and step to the GPU.  Here
// the games live in HBM: one 64-lane wave owns one game, its board sits in LDS, and a step is one launch that applies
// the sampled actions, decides termination, restarts finished games and writes the next observations (fp32 NCHW, the
// reference's layout, katago_observation.rs:41-92) and legal masks (bool rows and, optionally, the packed rows of the
// device rollout store) straight into the tensors the policy forward reads.
//
// Legality is decided differently from the reference (game.rs:262-335 makes every pseudo-legal move, recomputes the whole
// attack map and looks at the king): lanes enumerate the action space itself -- (square, direction) rays, knight jumps,
// drops, in the mover's perspective, so the action index falls out of the enumeration (spatial_action_mapper.rs:138-186)
// -- and every candidate is tested by looking OUTWARD from the king over a two-square overlay of the board in LDS.  The
// results are the same set of moves; tests/test_hip_shogi_env.py holds the kernels to the oracle bit for bit.
// Repetition uses a 64-bit mixed key of (board, hands, side) kept per ply, where the reference keeps Zobrist keys.
#include "common.h"

namespace {

constexpr int kTypes = 139, kBoardMoves = 81 * 80 * 2;
// action spaces: spatial 81 x 139 = 11 259 (spatial_action_mapper.rs), default 81 x 80 x 2 + 81 x 7 = 13 527 (action_mapper.rs:17-19);
// observation planes: katago 50 (katago_observation.rs), default 46 (observation.rs: planes 44-45 reserved)
constexpr int kMaxWords = (13527 + 31) / 32, kMaxCand = 1024;
__host__ __device__ constexpr int action_space(int amode) { return amode ? 81 * kTypes : kBoardMoves + 81 * 7; }
constexpr int kStateBytes = 128;      // board[81] hands[14] side in_check pad[3] | ply u32 @100 | key u64 @104 | reps u32 @112
enum { PAWN = 1, LANCE, KNIGHT, SILVER, GOLD, BISHOP, ROOK, KING };
constexpr int WHITE_BIT = 0x10, PROM_BIT = 0x20;
enum { R_PROGRESS = 0, R_CHECKMATE, R_REPETITION, R_PERPETUAL, R_IMPASSE, R_MAXMOVES };   // step_result.rs:9-16

// directions: N NE E SE S SW W NW (spatial_action_mapper.rs:31-40); "N" is toward row 0
// (row, column) steps as 2-bit fields of two immediates: the tables are on dependent-load chains of the ray walks
__device__ __forceinline__ int dir_dr(int d) { return ((0x1A90 >> (2 * d)) & 3) - 1; }      // -1 -1 0 1 1 1 0 -1
__device__ __forceinline__ int dir_dc(int d) { return ((0x01A9 >> (2 * d)) & 3) - 1; }      //  0  1 1 1 0 -1 -1 -1

// step (low byte) and slide (high byte) direction sets of every piece byte, in board directions (attack.rs:56-113)
__host__ __device__ constexpr unsigned dirs_of(int pc) {
    const unsigned N = 1, NE = 2, E = 4, SE = 8, S = 16, SW = 32, W = 64, NW = 128, gold = N | NE | NW | E | W | S;
    const int t = pc & 15;
    const bool pr = pc & PROM_BIT, wh = pc & WHITE_BIT;
    unsigned st = 0, sl = 0;
    if (pr) {
        if (t == PAWN || t == LANCE || t == KNIGHT || t == SILVER) st = gold;
        else if (t == BISHOP) { st = N | E | S | W; sl = NE | SE | SW | NW; }
        else if (t == ROOK) { st = NE | SE | SW | NW; sl = N | E | S | W; }
    } else {
        if (t == PAWN) st = N;
        else if (t == LANCE) sl = N;
        else if (t == SILVER) st = N | NE | NW | SE | SW;
        else if (t == GOLD) st = gold;
        else if (t == BISHOP) sl = NE | SE | SW | NW;
        else if (t == ROOK) sl = N | E | S | W;
        else if (t == KING) st = 255;
    }
    if (wh) { st = ((st << 4) | (st >> 4)) & 255; sl = ((sl << 4) | (sl >> 4)) & 255; }
    return st | (sl << 8);
}

struct EnvArgs {
    uint8_t* state; unsigned long long* keys; uint8_t* checks;
    const long long* actions; unsigned long long* err;     // err[0]: this step's refusal, err[1]: the latch (see ka_shogi_env_step)
    float* obs; uint8_t* mask; uint32_t* mask_bits;
    float* rewards; uint8_t* terminated; uint8_t* truncated; float* terminal_obs; uint8_t* current_players;
    uint8_t* captured; uint8_t* term_reason; uint16_t* ply_out; int* material; unsigned long long* stats;
    int amode, obs_ch;               // action mode 1 spatial / 0 default; observation planes 50 (katago) / 46 (default)
    int n, max_ply, mode;            // mode 0 reset, 1 step, 2 refresh (derive everything from board / hands / side as placed)
};

// the board in LDS (explicit address space: these helpers are not always inlined) with up to two squares replaced
typedef const __attribute__((address_space(3))) uint8_t* lds_board;
typedef const __attribute__((address_space(3))) uint16_t* lds_dirs;      // dirs_of() of every piece byte, in LDS
struct View { lds_board b; lds_dirs dirs; int o1, p1, o2, p2; };
__device__ __forceinline__ int at(const View& v, int sq) { return sq == v.o1 ? v.p1 : sq == v.o2 ? v.p2 : v.b[sq]; }

// is `sq` attacked by a piece of colour `by`?  Looks outward from the square: the first piece met along each of the
// eight lines attacks it if it steps (distance 1) or slides back along that line; plus the two knight origins.
// One probe = one line (d < 8) or one knight origin (d = 8, 9), so that ten lanes can share a test.
__device__ __forceinline__ bool attacked_from(const View& v, int sq, int by, int d) {
    const int r = sq / 9, c = sq % 9;
    if (d < 8) {
        const int dr = dir_dr(d), dc = dir_dc(d);
        const unsigned need = 1u << ((d + 4) & 7);
        int rr = r + dr, cc = c + dc, k = 1;
        while ((unsigned)rr < 9u && (unsigned)cc < 9u) {
            const int p = at(v, rr * 9 + cc);
            if (p) {
                if (((p >> 4) & 1) != by) return false;
                const unsigned m = v.dirs[p & 63];
                return (k == 1 && (m & need)) || ((m >> 8) & need);
            }
            rr += dr; cc += dc; ++k;
        }
        return false;
    }
    const int kr = by ? r - 2 : r + 2, kc = c + (d == 8 ? -1 : 1);
    return (unsigned)kr < 9u && (unsigned)kc < 9u && at(v, kr * 9 + kc) == (KNIGHT | (by ? WHITE_BIT : 0));
}
__device__ bool attacked(const View& v, int sq, int by) {
#pragma unroll 1
    for (int d = 0; d < 10; ++d) if (attacked_from(v, sq, by, d)) return true;
    return false;
}

// does the piece `pc` standing on `from` attack `target`? (rules.rs:136-176)
__device__ bool attacks(const View& v, int from, int pc, int target) {
    const int fr = from / 9, fc = from % 9, tr = target / 9, tc = target % 9;
    const int dr = tr - fr, dc = tc - fc;
    if ((pc & 15) == KNIGHT && !(pc & PROM_BIT)) return dr == ((pc & WHITE_BIT) ? 2 : -2) && (dc == 1 || dc == -1);
    const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
    if ((adr | adc) == 0 || !(dr == 0 || dc == 0 || adr == adc)) return false;
    const int ur = (dr > 0) - (dr < 0), uc = (dc > 0) - (dc < 0), dist = adr > adc ? adr : adc;
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (dir_dr(i) == ur && dir_dc(i) == uc) d = i;
    const unsigned m = v.dirs[pc & 63], bit = 1u << d;
    if (dist == 1 && (m & bit)) return true;
    if (!((m >> 8) & bit)) return false;
    for (int k = 1; k < dist; ++k) if (at(v, (fr + ur * k) * 9 + fc + uc * k)) return false;
    return true;
}

// rules.rs:18-131: would a pawn of `me` dropped on `to` leave the other king attacked with no way out?
__device__ bool pawn_drop_mates(lds_board board, lds_dirs dirs, int to, int me, int opp_king) {
    if (opp_king < 0) return false;
    const int opp = me ^ 1, pawn = PAWN | (me ? WHITE_BIT : 0);
    const View v{board, dirs, to, pawn, -1, 0};
    if (!attacked(v, opp_king, me)) return false;
    const int kr = opp_king / 9, kc = opp_king % 9;
    for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {          // the king steps aside or takes
        if (!dr && !dc) continue;
        const int r = kr + dr, c = kc + dc;
        if ((unsigned)r >= 9u || (unsigned)c >= 9u) continue;
        const int q = at(v, r * 9 + c);
        if (q && ((q >> 4) & 1) == opp) continue;
        if (attacked(v, r * 9 + c, me)) continue;
        return false;
    }
    for (int sq = 0; sq < 81; ++sq) {                                             // another piece takes the pawn
        const int pc = at(v, sq);
        if (!pc || ((pc >> 4) & 1) != opp || (pc & 15) == KING) continue;
        if (!attacks(v, sq, pc, to)) continue;
        const View w{board, dirs, sq, 0, to, pc};
        if (!attacked(w, opp_king, me)) return false;
    }
    return true;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned long long wave_xor64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo ^= __shfl_xor(lo, o); hi ^= __shfl_xor(hi, o); }
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ int start_piece(int sq) {          // position.rs:45-93
    const int r = sq / 9, c = sq % 9;
    const int back = c == 0 || c == 8 ? LANCE : c == 1 || c == 7 ? KNIGHT : c == 2 || c == 6 ? SILVER : c == 3 || c == 5 ? GOLD : KING;
    if (r == 0) return back | WHITE_BIT;
    if (r == 8) return back;
    if (r == 2) return PAWN | WHITE_BIT;
    if (r == 6) return PAWN;
    if (r == 1) return c == 1 ? (ROOK | WHITE_BIT) : c == 7 ? (BISHOP | WHITE_BIT) : 0;
    if (r == 7) return c == 1 ? BISHOP : c == 7 ? ROOK : 0;
    return 0;
}

__device__ __forceinline__ int piece_value(int t, bool pr) {  // rules.rs:333-350
    const int plain[9] = {0, 1, 3, 4, 5, 6, 8, 10, 0}, prom[9] = {0, 7, 6, 6, 6, 6, 10, 12, 0};
    return pr ? prom[t] : plain[t];
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void shogi_env_kernel(EnvArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_board[96];
    __shared__ uint8_t s_hands[16];
    __shared__ uint16_t s_dirs[64];
    __shared__ uint32_t s_bits[kMaxWords + 2];
    __shared__ uint32_t s_cand[kMaxCand];      // pseudo-legal candidates: at most 593 legal moves exist in any position; appends are clamped
    __shared__ int s_ncand, s_np, s_nh;
    __shared__ uint8_t s_plist[48], s_hlist[8];
    __shared__ float s_plane[22];

    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= a.n) return;
    const int kA = action_space(a.amode), kWords = (kA + 31) >> 5, kObs = a.obs_ch * 81;
    // an action of this step was refused: nothing moves in ANY game (vec_env.rs:651-690).  The outputs of this step are then
    // written again from the unchanged positions (zero rewards, no flags), so that the buffers the caller flips to hold the
    // positions to move and their masks -- the next step validates against them -- also when the caller reads the flag late.
    const bool hold = a.mode == 1 && a.err[0] != 0;
    // the refusal is latched in a second word that no launch clears: a caller that reads the flag late still sees the FIRST
    // refused step (and the action it named) after any number of further steps
    if (hold && env == 0 && lane == 0 && a.err[1] == 0) a.err[1] = a.err[0];
    uint8_t* st = a.state + (size_t)env * kStateBytes;
    unsigned long long* keys = a.keys + (size_t)env * (a.max_ply > 0 ? a.max_ply : 1);
    uint8_t* checks = a.checks + (size_t)env * (a.max_ply > 0 ? a.max_ply : 1);

    const lds_board brd = (lds_board)s_board;
    const lds_dirs drs = (lds_dirs)s_dirs;
    s_dirs[lane] = (uint16_t)dirs_of(lane);
    int side, ply, reps, in_check;
    unsigned long long key;
    auto load_board = [&]() {
        for (int i = lane; i < 81; i += 64) s_board[i] = st[i];
        if (lane < 14) s_hands[lane] = st[81 + lane];
        __syncthreads();
    };
    auto set_start = [&]() {
        for (int i = lane; i < 81; i += 64) s_board[i] = (uint8_t)start_piece(i);
        if (lane < 14) s_hands[lane] = 0;
        __syncthreads();
    };
    auto king_of = [&](int color) {                            // position.rs:141-150 (-1: none)
        const int target = KING | (color ? WHITE_BIT : 0);
        int k = 127;
        for (int i = lane; i < 81; i += 64) if (s_board[i] == target) k = min(k, i);
        k = wave_min_i(k);
        return k == 127 ? -1 : k;
    };
    auto position_key = [&]() {
        unsigned long long h = 0;
        for (int i = lane; i < 81; i += 64) { const int p = s_board[i]; if (p) h ^= mix64((unsigned long long)(i * 64 + p)); }
        if (lane < 14 && s_hands[lane]) h ^= mix64(0x4000ull + lane * 32 + s_hands[lane]);
        h = wave_xor64(h);
        return side ? h ^ mix64(0x8000ull) : h;
    };
    auto side_in_check = [&](int color) {                      // game.rs:98-105 (wave-uniform result); ten lanes, one probe each
        const int k = king_of(color);
        bool hit = false;
        if (k >= 0 && lane < 10) { const View v{brd, drs, -1, 0, -1, 0}; hit = attacked_from(v, k, color ^ 1, lane); }
        return (int)(__ballot(hit) != 0);
    };

    int terminal = R_PROGRESS, winner = -1, last_mover = 0, cap = 0;
    if (a.mode == 0) {
        set_start(); side = 0; ply = 0; reps = 1; in_check = 0; key = position_key();
    } else if (a.mode == 2) {
        load_board(); side = st[95]; ply = 0; reps = 1; in_check = side_in_check(side); key = position_key();
    } else if (hold) {
        load_board();
        side = st[95]; in_check = st[96];
        ply = *reinterpret_cast<const uint32_t*>(st + 100);
        key = *reinterpret_cast<const unsigned long long*>(st + 104);
        reps = (int)*reinterpret_cast<const uint32_t*>(st + 112);
        last_mover = side ^ 1;
    } else {
        load_board();
        side = st[95]; in_check = st[96];
        ply = *reinterpret_cast<const uint32_t*>(st + 100);
        key = *reinterpret_cast<const unsigned long long*>(st + 104);
        // ---- make_move (game.rs:107-188); the position before the move and whether its mover stood in check are history
        if (lane == 0) { keys[ply] = key; checks[ply] = (uint8_t)in_check; }
        const int act = (int)a.actions[env];
        if (lane == 0) {
            // decode in the mover's perspective (spatial_action_mapper.rs:188-279 / action_mapper.rs:79-110)
            int from_p, to_p, promote = 0, drop = -1;
            if (a.amode) {
                const int slot = act % kTypes;
                from_p = act / kTypes;
                if (slot < 128) {
                    promote = slot >= 64;
                    const int b = slot & 63, d = b >> 3, dist = (b & 7) + 1;
                    to_p = (from_p / 9 + dir_dr(d) * dist) * 9 + from_p % 9 + dir_dc(d) * dist;
                } else if (slot < 132) {
                    const int k = slot - 128;
                    promote = k & 1; to_p = (from_p / 9 - 2) * 9 + from_p % 9 + ((k >> 1) ? 1 : -1);
                } else { drop = slot - 132; to_p = from_p; }
            } else if (act < kBoardMoves) {
                from_p = act / 160;
                const int rem = act % 160, off = rem >> 1;
                promote = rem & 1; to_p = off >= from_p ? off + 1 : off;
            } else { drop = (act - kBoardMoves) % 7; to_p = from_p = (act - kBoardMoves) / 7; }
            const int to = side ? 80 - to_p : to_p;
            if (drop < 0) {
                const int real = side ? 80 - from_p : from_p;
                const int pc = s_board[real];
                cap = s_board[to];
                if (cap) s_hands[side * 7 + (cap & 15) - 1]++;
                s_board[real] = 0;
                s_board[to] = (uint8_t)(promote ? pc | PROM_BIT : pc);
            } else {
                s_hands[side * 7 + drop]--;
                s_board[to] = (uint8_t)((drop + 1) | (side ? WHITE_BIT : 0));
            }
        }
        cap = __shfl(cap, 0);
        __syncthreads();
        last_mover = side; side ^= 1; ++ply;
        key = position_key();
        in_check = side_in_check(side);
        // ---- check_termination (game.rs:355-387): move limit, repetition, impasse; "no legal move" follows below
        int matches = 0, unchecked = 0;
        for (int j = lane; j < ply; j += 64) if (keys[j] == key) { ++matches; unchecked += !checks[j]; }
        matches = wave_sum_i(matches); unchecked = wave_sum_i(unchecked);
        reps = 1 + matches;
        if (ply >= a.max_ply) terminal = R_MAXMOVES;
        else if (reps >= 4) {                                  // rules.rs:190-235
            if (matches > 0 && unchecked == 0) { terminal = R_PERPETUAL; winner = side; }
            else terminal = R_REPETITION;
        } else {                                               // rules.rs:228-262
            const int bk = king_of(0), wk = king_of(1);
            if (bk >= 0 && wk >= 0 && bk / 9 <= 2 && wk / 9 >= 6) {
                int acc = 0;                                   // bytes: zone count black, white; points black, white
                for (int i = lane; i < 81; i += 64) {
                    const int p = s_board[i];
                    if (!p) continue;
                    const int c = (p >> 4) & 1, t = p & 15, r = i / 9;
                    const int pts = t == ROOK || t == BISHOP ? 5 : t == KING ? 0 : 1;
                    const int zone = c ? r >= 6 : r <= 2;
                    acc += (zone << (8 * c)) + (pts << (16 + 8 * c));
                }
                if (lane < 14) acc += (s_hands[lane] * (lane % 7 >= 5 ? 5 : 1)) << (16 + 8 * (lane / 7));
                acc = wave_sum_i(acc);
                const int bz = acc & 255, wz = (acc >> 8) & 255, bs = (acc >> 16) & 255, ws = (acc >> 24) & 255;
                if (bz >= 10 && wz >= 10) {
                    if (bs >= 24 && ws >= 24) { terminal = R_IMPASSE; winner = -1; }
                    else if (bs >= 24) { terminal = R_IMPASSE; winner = 0; }
                    else if (ws >= 24) { terminal = R_IMPASSE; winner = 1; }
                }
            }
        }
    }

    // material balance of the position after the move, seen by the mover (rules.rs:356-383; every step, vec_env.rs:371-374)
    int bal = 0;
    if (a.mode == 1) {
        for (int i = lane; i < 81; i += 64) {
            const int p = s_board[i];
            if (!p || (p & 15) == KING) continue;
            const int v = piece_value(p & 15, p & PROM_BIT);
            bal += ((p >> 4) & 1) == last_mover ? v : -v;
        }
        if (lane < 7) bal += piece_value(lane + 1, false) * ((int)s_hands[last_mover * 7 + lane] - (int)s_hands[(last_mover ^ 1) * 7 + lane]);
        bal = wave_sum_i(bal);
    }

    // observation writer (observation.rs:81-153, katago_observation.rs:41-92): 28 piece planes from the board, 22 constant planes
    auto write_obs = [&](float* out) {
        if (lane < 14) {
            const float mx = lane % 7 == 0 ? 18.f : lane % 7 >= 5 ? 2.f : 4.f;
            const int who = lane < 7 ? side : side ^ 1;
            s_plane[lane] = __fdiv_rn((float)s_hands[who * 7 + lane % 7], mx);
        } else if (lane == 14) s_plane[14] = side == 0 ? 1.f : 0.f;
        else if (lane == 15) s_plane[15] = a.max_ply == 0 ? 0.f : fminf(fmaxf(__fdiv_rn((float)ply, (float)a.max_ply), 0.f), 1.f);
        else if (lane < 20) { const int prior = reps - 1, ch = lane - 16; s_plane[lane] = (a.obs_ch == 50 && (ch < 3 ? prior == ch + 1 : prior >= 4)) ? 1.f : 0.f; }
        else if (lane == 20) s_plane[20] = (a.obs_ch == 50 && in_check) ? 1.f : 0.f;
        else if (lane == 21) s_plane[21] = 0.f;
        __syncthreads();
        // 28 piece planes: zeros, then one 1.0 per piece; the other planes are constants.  (Computing every element from
        // the board cost about a third of the kernel's instructions.)
        f32x2* o2 = reinterpret_cast<f32x2*>(out);
        constexpr int kPiecePairs = 28 * 81 / 2;
        for (int i = lane; i < kPiecePairs; i += 64) o2[i] = f32x2{0.f, 0.f};
        for (int i = kPiecePairs + lane; i < kObs / 2; i += 64) {
            const int idx = 2 * i;
            o2[i] = f32x2{s_plane[idx / 81 - 28], s_plane[(idx + 1) / 81 - 28]};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zeros are in L2 before other lanes' ones follow them
        for (int sq = lane; sq < 81; sq += 64) {
            const int p = s_board[side ? 80 - sq : sq];
            if (p) {
                const int t = p & 15, own = ((p >> 4) & 1) == side;
                const int pch = (p & PROM_BIT) ? (own ? 8 : 22) + (t <= SILVER ? t - 1 : t - 2) : (own ? 0 : 14) + t - 1;
                out[pch * 81 + sq] = 1.f;
            }
        }
        __syncthreads();
    };


    // ---- legal moves of the side to move into s_bits.  Pass 0 may find none (= mate); then, as after every other
    //      ending, the finished game is reported, a fresh one set up, and pass 1 generates the start position's moves.
    int result = R_PROGRESS, final_ply = ply;
    for (int pass = 0; pass < 2; ++pass) {
        if (terminal == R_PROGRESS) {
            for (int i = lane; i < kWords + 2; i += 64) s_bits[i] = 0;
            if (lane == 0) { s_ncand = 0; s_np = 0; s_nh = 0; }
            __syncthreads();
            const int me = side, mine = me ? WHITE_BIT : 0;
            // candidates in the mover's perspective (movegen.rs:112-203): the mover's pieces are compacted first, then one
            // task per (piece, direction or knight side) and one per (hand type held, square)
            for (int sq = lane; sq < 81; sq += 64) {
                const int p = s_board[sq];
                if (p && (p & WHITE_BIT) == mine) s_plist[atomicAdd(&s_np, 1)] = (uint8_t)sq;
            }
            if (lane < 7 && s_hands[me * 7 + lane]) s_hlist[atomicAdd(&s_nh, 1)] = (uint8_t)lane;
            __syncthreads();
            const int nboard = s_np * 10, ntask = nboard + s_nh * 81;
            for (int task = lane; task < ntask; task += 64) {
                if (task < nboard) {
                    const int from = s_plist[task / 10], d_p = task % 10;
                    const int sq_p = me ? 80 - from : from;
                    const int p = s_board[from];
                    if (d_p < 8) {
                        const int d = me ? (d_p + 4) & 7 : d_p;
                        const unsigned m = s_dirs[p & 63];
                        const bool slide = (m >> 8) & (1u << d);
                        if (!slide && !(m & (1u << d))) continue;
                        const int t = p & 15;
                        const bool can = !(p & PROM_BIT) && t != GOLD && t != KING;
                        const int fr_p = sq_p / 9, dr = dir_dr(d), dc = dir_dc(d), drp = dir_dr(d_p);
                        int rr = from / 9 + dr, cc = from % 9 + dc;
                        for (int k = 1; k <= (slide ? 8 : 1) && (unsigned)rr < 9u && (unsigned)cc < 9u; ++k, rr += dr, cc += dc) {
                            const int to = rr * 9 + cc, q = s_board[to];
                            if (q && (q & WHITE_BIT) == mine) break;
                            const int to_row_p = fr_p + drp * k;
                            const bool must = can && (t == PAWN || t == LANCE) && to_row_p == 0;      // movegen.rs:33-45
                            const bool opt = can && !must && (fr_p <= 2 || to_row_p <= 2);
                            const unsigned base = (unsigned)from | ((unsigned)to << 7);
                            const int to_p = me ? 80 - to : to;
                            const int act = a.amode ? sq_p * kTypes + d_p * 8 + k - 1 : sq_p * 160 + (to_p > sq_p ? to_p - 1 : to_p) * 2;
                            const int pstep = a.amode ? 64 : 1;
                            if (!must) s_cand[min(atomicAdd(&s_ncand, 1), kMaxCand - 1)] = base | ((unsigned)act << 18);
                            if (must || opt) s_cand[min(atomicAdd(&s_ncand, 1), kMaxCand - 1)] = base | (1u << 14) | ((unsigned)(act + pstep) << 18);
                            if (q) break;
                        }
                    } else {
                        if (p != (KNIGHT | mine)) continue;
                        const int sd = d_p - 8;
                        const int tr_p = sq_p / 9 - 2, tc_p = sq_p % 9 + (sd ? 1 : -1);
                        if (tr_p < 0 || (unsigned)tc_p >= 9u) continue;
                        const int to_p = tr_p * 9 + tc_p, to = me ? 80 - to_p : to_p, q = s_board[to];
                        if (q && (q & WHITE_BIT) == mine) continue;
                        const bool must = tr_p <= 1, opt = !must && tr_p <= 2;
                        const unsigned base = (unsigned)from | ((unsigned)to << 7);
                        const int act = a.amode ? sq_p * kTypes + 128 + sd * 2 : sq_p * 160 + (to_p > sq_p ? to_p - 1 : to_p) * 2;
                        if (!must) s_cand[min(atomicAdd(&s_ncand, 1), kMaxCand - 1)] = base | ((unsigned)act << 18);
                        if (must || opt) s_cand[min(atomicAdd(&s_ncand, 1), kMaxCand - 1)] = base | (1u << 14) | ((unsigned)(act + 1) << 18);
                    }
                } else {
                    const int u = task - nboard, h = s_hlist[u / 81], sq_p = u % 81;
                    const int to = me ? 80 - sq_p : sq_p;
                    if (s_board[to]) continue;
                    const int row_p = sq_p / 9;
                    if ((h <= 1 && row_p == 0) || (h == 2 && row_p <= 1)) continue;               // movegen.rs:50-62
                    if (h == 0) {                                                                 // game.rs:24-34, 274-277
                        bool nifu = false;
                        for (int r = 0; r < 9; ++r) nifu |= s_board[r * 9 + to % 9] == (PAWN | mine);
                        if (nifu) continue;
                    }
                    s_cand[min(atomicAdd(&s_ncand, 1), kMaxCand - 1)] = ((unsigned)to << 7) | ((unsigned)(h + 1) << 15) |
                                                     ((unsigned)(a.amode ? sq_p * kTypes + 132 + h : kBoardMoves + sq_p * 7 + h) << 18);
                }
            }
            __syncthreads();
            const int my_king = king_of(me), opp_king = king_of(me ^ 1);
            const int nc = min(s_ncand, kMaxCand);
            // King safety.  A move can only uncover the king when the moving piece stands on one of the king's eight
            // lines, and a drop never does; so unless the mover is in check already, only king moves and moves of aligned
            // pieces take the full test (the moved piece gone from `from`, present on `to`, look outward from the king).
            const int kr = my_king / 9, kc = my_king % 9;
            for (int i = lane; i < nc; i += 64) {
                const unsigned c = s_cand[i];
                const int from = c & 127, to = (c >> 7) & 127, promote = (c >> 14) & 1, drop = (c >> 15) & 7, act = c >> 18;
                bool ok = my_king >= 0;
                if (drop) {
                    if (ok && in_check) { const View v{brd, drs, to, drop | mine, -1, 0}; ok = !attacked(v, my_king, me ^ 1); }
                    if (ok && drop == PAWN) ok = !pawn_drop_mates(brd, drs, to, me, opp_king);
                } else if (ok) {
                    const int pc = s_board[from];
                    const bool king = (pc & 15) == KING;
                    const int dr = from / 9 - kr, dc = from % 9 - kc;
                    const bool aligned = dr == 0 || dc == 0 || dr == dc || dr == -dc;
                    if (king || in_check || aligned) {
                        const View v{brd, drs, from, 0, to, promote ? pc | PROM_BIT : pc};
                        ok = !attacked(v, king ? to : my_king, me ^ 1);
                    }
                }
                if (ok) atomicOr(&s_bits[act >> 5], 1u << (act & 31));
            }
            __syncthreads();
            unsigned any = 0;
            for (int w = lane; w < kWords; w += 64) any |= s_bits[w];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o);
            if (pass == 0 && a.mode == 1 && !hold && any == 0) { terminal = R_CHECKMATE; winner = side ^ 1; }   // game.rs:374-386
        }
        if (terminal == R_PROGRESS) break;
        // the game ended with this move (vec_env.rs:395-423)
        result = terminal;
        if (lane == 0) {
            atomicAdd(&a.stats[0], 1ull);
            atomicAdd(&a.stats[3], (unsigned long long)ply);
            if (terminal == R_REPETITION || (terminal == R_IMPASSE && winner < 0)) atomicAdd(&a.stats[1], 1ull);
            if (terminal == R_MAXMOVES) atomicAdd(&a.stats[2], 1ull);
        }
        write_obs(a.terminal_obs + (size_t)env * kObs);        // the finished game, seen by its side to move
        set_start(); side = 0; ply = 0; reps = 1; in_check = 0; key = position_key();
        terminal = R_PROGRESS;
    }

    if (a.mode == 1 && lane == 0) {                            // per-step scalars (vec_env.rs:358-390)
        const bool trunc = result == R_MAXMOVES, term = result != R_PROGRESS && !trunc;
        a.terminated[env] = term; a.truncated[env] = trunc;
        float rw = 0.f;                                        // vec_env.rs:98-124
        if (result == R_CHECKMATE || result == R_PERPETUAL || (result == R_IMPASSE && winner >= 0)) rw = winner == last_mover ? 1.f : -1.f;
        a.rewards[env] = rw;
        a.term_reason[env] = (uint8_t)result;
        a.ply_out[env] = (uint16_t)final_ply;
        a.material[env] = bal;
        a.captured[env] = cap ? (uint8_t)((cap & 15) - 1) : 255;
    }

    // ---- state back to HBM, observation, masks
    if (lane == 0) {
        st[95] = (uint8_t)side; st[96] = (uint8_t)in_check;
        *reinterpret_cast<uint32_t*>(st + 100) = (uint32_t)ply;
        *reinterpret_cast<unsigned long long*>(st + 104) = key;
        *reinterpret_cast<uint32_t*>(st + 112) = (uint32_t)reps;
        if (a.current_players) a.current_players[env] = (uint8_t)side;
    }
    for (int i = lane; i < 81; i += 64) st[i] = s_board[i];
    if (lane < 14) st[81 + lane] = s_hands[lane];
    write_obs(a.obs + (size_t)env * kObs);

    if (a.mask_bits) for (int i = lane; i < kWords; i += 64) a.mask_bits[(size_t)env * kWords + i] = s_bits[i];
    if (a.mask) {
        // bool rows of 11 259 bytes start at odd addresses: single bytes up to the first 16-byte boundary, 16-byte
        // pieces (16 mask bits spread over four words) in the middle, single bytes at the end
        uint8_t* row = a.mask + (size_t)env * kA;
        const int head = (int)((16 - (reinterpret_cast<uintptr_t>(row) & 15)) & 15);
        const int body = (kA - head) >> 4, tail0 = head + body * 16;
        auto bits16 = [&](int b) {
            const unsigned long long w = ((unsigned long long)s_bits[(b >> 5) + 1] << 32) | s_bits[b >> 5];
            return (unsigned)(w >> (b & 31)) & 0xFFFFu;
        };
        if (lane < head) row[lane] = (s_bits[lane >> 5] >> (lane & 31)) & 1;
        for (int i = lane; i < body; i += 64) {
            const unsigned x = bits16(head + i * 16);
            uint4 v;
            v.x = ((x & 15) * 0x00204081u) & 0x01010101u;
            v.y = (((x >> 4) & 15) * 0x00204081u) & 0x01010101u;
            v.z = (((x >> 8) & 15) * 0x00204081u) & 0x01010101u;
            v.w = (((x >> 12) & 15) * 0x00204081u) & 0x01010101u;
            *reinterpret_cast<uint4*>(row + head + i * 16) = v;
        }
        for (int b = tail0 + lane; b < kA; b += 64) row[b] = (s_bits[b >> 5] >> (b & 31)) & 1;
    }
}

// an action is accepted when it is inside the action space and set in the mask handed out last (vec_env.rs:651-690);
// err = ((n - index of the first refused env) << 32) | the refused action (clamped to 32 bits), 0 when every action stands
__global__ void shogi_validate_kernel(const long long* actions, const uint8_t* mask, const uint32_t* bits, int n, int kA, unsigned long long* err) {
    const int kWords = (kA + 31) >> 5;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long act = actions[i];
    bool ok = act >= 0 && act < kA;
    if (ok) ok = mask ? mask[(size_t)i * kA + act] != 0 : (bits[(size_t)i * kWords + (act >> 5)] >> (act & 31)) & 1;
    if (!ok) {
        const long long c = act < -2147483647LL - 1 ? -2147483647LL - 1 : (act > 2147483647LL ? 2147483647LL : act);
        atomicMax(err, ((unsigned long long)(n - i) << 32) | (unsigned int)(int)c);
    }
}

}  // namespace

// ---------------------------------------------------------------- C ABI (include/keisei_amd.h)
extern "C" int ka_shogi_env_state_bytes(void) { return kStateBytes; }

static int launch_env(EnvArgs a, hipStream_t st) {
    hipLaunchKernelGGL(shogi_env_kernel, dim3(a.n), dim3(64), 0, st, a);
    return ka_check_launch("shogi_env");
}

extern "C" int ka_shogi_env_action_space(int action_mode) { return action_space(action_mode); }

extern "C" int ka_shogi_env_reset(void* state, void* keys, void* checks, int n, int max_ply, int obs_mode, int action_mode,
                                  float* obs, void* mask, void* mask_bits, void* current_players, int refresh, void* stream) {
    KA_REQUIRE(state && keys && checks && obs && n > 0 && max_ply >= 0 && (mask || mask_bits), "shogi_env_reset: bad arguments");
    KA_REQUIRE((obs_mode == 0 || obs_mode == 1) && (action_mode == 0 || action_mode == 1), "shogi_env_reset: modes are 0 (default) or 1 (katago / spatial)");
    EnvArgs a{};
    a.state = static_cast<uint8_t*>(state); a.keys = static_cast<unsigned long long*>(keys); a.checks = static_cast<uint8_t*>(checks);
    a.obs = obs; a.mask = static_cast<uint8_t*>(mask); a.mask_bits = static_cast<uint32_t*>(mask_bits);
    a.current_players = static_cast<uint8_t*>(current_players);
    a.amode = action_mode; a.obs_ch = obs_mode ? 50 : 46;
    a.n = n; a.max_ply = max_ply; a.mode = refresh ? 2 : 0;
    return launch_env(a, static_cast<hipStream_t>(stream));
}

extern "C" int ka_shogi_env_step(void* state, void* keys, void* checks, const long long* actions, int n, int max_ply,
                                 int obs_mode, int action_mode, const void* prev_mask, const void* prev_mask_bits, int* err,
                                 float* obs, void* mask, void* mask_bits, float* rewards, void* terminated, void* truncated,
                                 float* terminal_obs, void* current_players, void* captured, void* term_reason,
                                 void* ply_count, int* material, void* stats, void* stream) {
    KA_REQUIRE(state && keys && checks && actions && err && obs && rewards && terminated && truncated && terminal_obs &&
               current_players && captured && term_reason && ply_count && material && stats && n > 0 && max_ply >= 0,
               "shogi_env_step: bad arguments");
    KA_REQUIRE((mask || mask_bits) && (prev_mask || prev_mask_bits), "shogi_env_step: needs the bool or the packed masks");
    KA_REQUIRE((obs_mode == 0 || obs_mode == 1) && (action_mode == 0 || action_mode == 1), "shogi_env_step: modes are 0 (default) or 1 (katago / spatial)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(err, 0, sizeof(unsigned long long), st) != hipSuccess) { ka_set_error("shogi_env_step: memset failed"); return KA_ERR_HIP; }
    hipLaunchKernelGGL(shogi_validate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, actions,
                       static_cast<const uint8_t*>(prev_mask), static_cast<const uint32_t*>(prev_mask_bits), n, action_space(action_mode), reinterpret_cast<unsigned long long*>(err));
    EnvArgs a{};
    a.state = static_cast<uint8_t*>(state); a.keys = static_cast<unsigned long long*>(keys); a.checks = static_cast<uint8_t*>(checks);
    a.actions = actions; a.err = reinterpret_cast<unsigned long long*>(err);
    a.obs = obs; a.mask = static_cast<uint8_t*>(mask); a.mask_bits = static_cast<uint32_t*>(mask_bits);
    a.rewards = rewards; a.terminated = static_cast<uint8_t*>(terminated); a.truncated = static_cast<uint8_t*>(truncated);
    a.terminal_obs = terminal_obs; a.current_players = static_cast<uint8_t*>(current_players);
    a.captured = static_cast<uint8_t*>(captured); a.term_reason = static_cast<uint8_t*>(term_reason);
    a.ply_out = static_cast<uint16_t*>(ply_count); a.material = material; a.stats = static_cast<unsigned long long*>(stats);
    a.amode = action_mode; a.obs_ch = obs_mode ? 50 : 46;
    a.n = n; a.max_ply = max_ply; a.mode = 1;
    return launch_env(a, st);
}
