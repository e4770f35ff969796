// Weight gradient of the 3x3 board convolution on the CDNA4 matrix cores.
//
//   dW[n,c,tap] = sum_{b,p} dY[b,p,n] * X'[b, p+tap, c]
//
// A "TN" GEMM whose reduction axis (b,p) is the slow axis of both NHWC operands.  One
// workgroup (8 waves) owns a 128(n) x 64(c) x 9(tap) slab of dW in registers (144 accumulator
// VGPRs per lane) and walks its share of the boards: per board the dY tile [81->96 rows][128]
// and the zero-haloed X' tile [121 squares][64] are staged into LDS (global loads for board
// i+1 are in flight while board i is multiplied), and the 9 taps reuse both tiles -- the tap
// is again a constant LDS row offset on the B operand.  bf16 operands are read with the
// hardware transpose read (ds_read_b64_tr_b16) so no transposed copy of either tensor ever
// exists; the f32 (parity) path needs one element per lane (v_mfma_f32_16x16x4_f32) and reads
// with plain ds_read_b32.  The batch is split over workgroups (split-K); partial slabs are
// summed by wgrad_reduce_kernel in a fixed order, so the result is run-to-run deterministic.
//
// Replaces: autograd's conv2d weight backward for se_resnet.py:50,52,110.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int kTC = 64;      // c-tile width; the n-tile width TN (128 or 64) is a template parameter

struct WgradArgs {
    const void* dy;      // (B,81,Cout)
    const void* x;       // (B,81,Cin)
    const float* in_scale;
    const float* in_shift;
    const float* in_bias;    // [B,Cin]
    float* slab;             // [nsplit][9][Cout][Cin]
    int B, Cin, Cout, relu, boards_per_split, ntn, ntiles, nsplit;
};

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

template <typename T, int TN> struct WG;
template <int TN> struct WG<bf16_t, TN> {
    // 128-wide tile (flat K): 81 rows + one zero row; the haloed X tile is 17 squares wide (as the conv3x3 image: a
    // board-row wrap then advances the square index by 9 = 1 mod 8, so the 8 consecutive rows of a transpose-read half
    // stay on 8 different 32-byte bank groups -- with the natural 11-wide tile two of them collide at every wrap;
    // rocprof: 30 % of the LDS cycles of this kernel were bank conflicts).  64-wide tile: per-board K, 81 -> 96 rows.
    static constexpr int KROWS = TN == 128 ? 82 : 96;
    static constexpr int PW = TN == 128 ? 17 : 11;   // padded board-row width of the X tile
    static constexpr int SY = TN * 2 + 32;           // dY tile row stride (bytes)
    static constexpr int SX = kTC * 2 + 32;
};
template <int TN> struct WG<float, TN> {
    static constexpr int PW = 11;
    static constexpr int KROWS = 84;                 // 81 -> 21 k-steps of 4
    static constexpr int SY = TN * 4 + 64;
    static constexpr int SX = kTC * 4 + 64;
};

// TN = 128: 512 threads, wave (nh, cq) = 64 output channels x 16 input channels; one workgroup per CU.
// TN = 64 : 256 threads, wave cq = all 64 output channels x 16 input channels; two independent workgroups per CU (the
//           staging / barrier phases of one run under the MFMAs of the other, and a workgroup can share a CU with a
//           workgroup of another kernel when the weight gradient overlaps the main stream).
// FUSED = the X operand gets the BatchNorm+ReLU+bias input transform (conv2's weight gradient); the plain form carries
// no per-channel coefficients.  (Ablation: the board-prefetch global loads cost 14 % of the kernel at their one-board
// distance; a second register set for a two-board distance does not fit -- it spills and is 30 % slower.)
template <typename T, int TN, bool FUSED>
__global__ __launch_bounds__(TN * 4, TN == 64 ? 2 : 1) void wgrad_kernel(WgradArgs a) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int NTHR = TN * 4;
    constexpr int ESZ = E::kSize, P16 = E::kPer16;
    constexpr int SY = WG<T, TN>::SY, SX = WG<T, TN>::SX, KROWS = WG<T, TN>::KROWS;
    constexpr int PW = WG<T, TN>::PW, XSQ = 11 * PW;             // haloed X tile: 11 rows of PW squares
    auto xsq = [](int p) { return (p / 9 + 1) * PW + (p % 9) + 1; };   // tile index of board square p
    constexpr int PY = TN * ESZ / 16, PX = kTC * ESZ / 16;      // 16-byte pieces per tile row
    constexpr int NY = (KA_BOARD * PY + NTHR - 1) / NTHR, NX = (KA_BOARD * PX + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // bf16: two tile sets (double buffer) -- a wave writes board b+1 into the other set while slower waves still
    // multiply board b, one barrier per board.  f32 tiles are twice as large: single set, two barriers per board.
    // FLAT (bf16, 128-wide tile): the K dimension is the FLAT row index over the workgroup's whole board range, cut
    // into steps of 32 rows that may straddle two boards (a ring of three board tiles), instead of 3 steps per board
    // with 15 zero rows in the last one: 81/32 = 2.53 steps per board, 15.6 % fewer MFMAs.
    constexpr bool FLAT = sizeof(T) == 2 && TN == 128;
    constexpr int NBUF = FLAT ? 3 : ((sizeof(T) == 2) ? 2 : 1);
    constexpr int TILE_BYTES = KROWS * SY + XSQ * SX;
    char* ytile = smem;
    char* xtile = smem + KROWS * SY;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nh = TN == 128 ? (wave & 1) : 0, cq = TN == 128 ? (wave >> 1) : wave;
    // XCD-aware workgroup -> (tile, split) map: workgroups are dealt round-robin over the 8 XCDs, so linear id L sits
    // on XCD L % 8.  All output tiles of one board range (split) are placed on the SAME XCD: they read the same dY / X
    // boards (dY is needed by every c-tile, X by every n-tile), which then hit that XCD's L2 instead of being fetched
    // 4x / 2x over the fabric (rocprof FETCH_SIZE before the remap: 5.7 activation tensors per launch, algorithmic 2).
    // Placement only affects speed, never correctness.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % a.ntiles;
    const int split = xcd + 8 * (slot / a.ntiles);
    if (split >= a.nsplit) return;               // padding workgroups when nsplit % 8 != 0 (uniform exit, no barrier yet)
    const int tn = tile % a.ntn, tc = tile / a.ntn;
    const int n0 = tn * TN, c0 = tc * kTC;
    const int bbeg = split * a.boards_per_split;
    const int bend = min(a.B, bbeg + a.boards_per_split);

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero both tiles once: dY pad rows (81..KROWS) and the X halo stay zero forever
    for (int i = tid; i < NBUF * TILE_BYTES / 16; i += NTHR)
        reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};

    // staging roles
    const int yj = tid % PY, xj = tid % PX;
    const bool ycol_ok = n0 + yj * P16 < a.Cout, xcol_ok = c0 + xj * P16 < a.Cin;
    constexpr int NSETS = 1;                              // board-prefetch register sets (two do not fit, see above)
    float sc[FUSED ? P16 : 1], sh[FUSED ? P16 : 1];
    const bool has_aff = FUSED && a.in_scale != nullptr;
    if (has_aff && xcol_ok) {
#pragma unroll
        for (int e = 0; e < P16; ++e) { sc[e] = a.in_scale[c0 + xj * P16 + e]; sh[e] = a.in_shift[c0 + xj * P16 + e]; }
    }
    vec16 ry[NSETS][NY], rx[NSETS][NX];
    float rb[FUSED ? P16 : 1];              // per-board bias of this thread's X channel piece
    auto zero16 = [&] { float z[P16];
#pragma unroll
        for (int e = 0; e < P16; ++e) z[e] = 0.f;
        return E::pack(z); };

    auto load_board = [&](int b, vec16 (&ry)[NY], vec16 (&rx)[NX]) {
        if (FUSED && a.in_bias && xcol_ok) {
#pragma unroll
            for (int e = 0; e < P16; ++e) rb[e] = a.in_bias[(size_t)b * a.Cin + c0 + xj * P16 + e];
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int row = (tid + i * NTHR) / PY;
            ry[i] = (row < KA_BOARD && ycol_ok)
                        ? *reinterpret_cast<const vec16*>(static_cast<const char*>(a.dy) +
                              ((size_t)(b * KA_BOARD + row) * a.Cout + n0 + yj * P16) * ESZ)
                        : zero16();
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = (tid + i * NTHR) / PX;
            rx[i] = (row < KA_BOARD && xcol_ok)
                        ? *reinterpret_cast<const vec16*>(static_cast<const char*>(a.x) +
                              ((size_t)(b * KA_BOARD + row) * a.Cin + c0 + xj * P16) * ESZ)
                        : zero16();
        }
    };
    auto store_board = [&](int b, const vec16 (&ry)[NY], const vec16 (&rx)[NX]) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int row = (tid + i * NTHR) / PY;
            if (row < KA_BOARD) *reinterpret_cast<vec16*>(ytile + row * SY + yj * 16) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = (tid + i * NTHR) / PX;
            if (row < KA_BOARD) {
                vec16 v = rx[i];
                if (FUSED && xcol_ok && (has_aff || a.relu || a.in_bias)) {
                    float f[P16];
                    E::unpack(v, f);
                    if (has_aff) {
#pragma unroll
                        for (int e = 0; e < P16; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < P16; ++e) f[e] = fmaxf(f[e], 0.f);
                    }
                    if (a.in_bias) {
#pragma unroll
                        for (int e = 0; e < P16; ++e) f[e] += rb[e];
                    }
                    v = E::pack(f);
                }
                *reinterpret_cast<vec16*>(xtile + xsq(row) * SX + xj * 16) = v;
            }
        }
    };

    // wave-uniform validity of this wave's tiles
    const int ntn_valid = min(4, max(0, (a.Cout - n0 - nh * 64 + 15) / 16));   // valid n-tiles of this wave
    const bool c_ok = c0 + cq * 16 < a.Cin;

    // Board j of the range travels through register set j % NSETS.  One set: the loads of board b+2 are issued in
    // iteration b, right after the set was emptied into the tile of board b+1.  Two sets: iteration b empties the set of
    // board b+1 and refills it with board b+3.
    constexpr int AHEAD = NSETS + 1;
    if (bbeg < bend) load_board(bbeg, ry[0], rx[0]);
    if (NBUF >= 2) {
        __syncthreads();                 // zero fill complete
        if (bbeg < bend) {
            store_board(bbeg, ry[0], rx[0]);
            if (bbeg + 1 < bend) load_board(bbeg + 1, ry[NSETS - 1], rx[NSETS - 1]);
            if (NSETS == 2 && bbeg + 2 < bend) load_board(bbeg + 2, ry[0], rx[0]);
        }
        __syncthreads();
    }
    auto stage_next = [&](int b, vec16 (&ys)[NY], vec16 (&xs)[NX]) {
        // stage board b+1 into the next tile of the ring (its registers were loaded AHEAD-1 iterations ago), then start
        // the loads of board b+AHEAD into the same set; the single barrier at the bottom closes both hazards
        if (b + 1 < bend) {
            ytile = smem + ((b - bbeg + 1) % NBUF) * TILE_BYTES; xtile = ytile + KROWS * SY;
            store_board(b + 1, ys, xs);
            if (b + AHEAD < bend) load_board(b + AHEAD, ys, xs);
        }
    };
    // one board; ys/xs = the register set of board b+1 (compile-time choice: the loop below is unrolled by two)
    auto board_iter = [&](int b, vec16 (&ys)[NY], vec16 (&xs)[NX]) {
        if (NBUF == 1) {
            __syncthreads();                 // previous board fully consumed
            store_board(b, ys, xs);
            __syncthreads();
            if (b + 1 < bend) load_board(b + 1, ys, xs);      // in flight during the MFMA phase
        } else {
            const int cur = (b - bbeg) % NBUF;
            stage_next(b, ys, xs);
            ytile = smem + cur * TILE_BYTES; xtile = ytile + KROWS * SY;
        }
        const bool skip = !c_ok || ntn_valid == 0;    // wave-uniform
        if (!skip) {

        if constexpr (sizeof(T) == 2) {
            // branch-free MFMA stream: tiles beyond Cout multiply zero-filled LDS columns and are never stored.
            // B fragments (the tap-shifted X rows) are fetched one tap ahead; the issue order is pinned so each
            // pair of transpose reads sits in front of the 4 MFMAs of the previous tap.
            // rows of k-step ks for this lane: k1 = 32 ks + 4q + (r>>2), k2 = k1 + 16 (k-slot permutation, see below).
            // Per board: ks = 0..2, rows >= 81 are the zero pad rows of the dY tile.  FLAT: ks counts 32-row steps of the
            // flat row index over the board range; a row belongs to board jl (this iteration's) or jl-1 (the previous
            // tile of the ring); rows past the range read the zero pad row 81.
            const int jl = b - bbeg, nb = bend - bbeg;
            const char* yprev = smem + ((jl + NBUF - 1) % NBUF) * TILE_BYTES;
            auto row_ptrs = [&](int k, const char*& yrow, const char*& xrow) {
                if (FLAT) {
                    int p = k - KA_BOARD * jl;
                    const bool prev = p < 0;
                    p = prev ? p + KA_BOARD : p;
                    const char* yt = prev ? yprev : ytile;
                    const bool pad = p >= KA_BOARD;                 // only past the end of the range (last step)
                    yrow = yt + (pad ? KA_BOARD : p) * SY;
                    xrow = yt + KROWS * SY + xsq(pad ? 0 : p) * SX;
                } else {
                    yrow = ytile + k * SY;
                    xrow = xtile + xsq(k < KA_BOARD ? k : 0) * SX;
                }
            };
            const int ks_lo = FLAT ? (KA_BOARD * jl) / 32 : 0;
            const int ks_hi = FLAT ? (jl + 1 == nb ? (KA_BOARD * nb + 31) / 32 : (KA_BOARD * (jl + 1)) / 32) : 3;
#pragma unroll 1
            for (int ks = ks_lo; ks < ks_hi; ++ks) {
                // MFMA k-slot (q, j) is mapped to tile row 4q+j (j<4) / 16+4q+(j-4) (j>=4) for BOTH operands (any
                // common permutation of k is legal): each 32-lane half of a transpose read then covers 8 CONSECUTIVE
                // rows, which the 32*odd-byte row strides spread over all 64 banks (the natural 8q+j map makes a half
                // read rows {k..k+3, k+8..k+11}: a guaranteed 2-way conflict)
                const int k1 = ks * 32 + 4 * q + (r >> 2), k2 = k1 + 16;
                const char *y1, *x1, *y2, *x2;
                row_ptrs(k1, y1, x1);
                row_ptrs(k2, y2, x2);
                const int colx = (cq * 16 + 4 * (r & 3)) * 2;
                // B fragments (the tap-shifted X rows) are fetched one tap ahead; the issue order is pinned so each pair
                // of transpose reads sits in front of the 4 MFMAs of the previous tap.  Branch-free MFMA stream: tiles
                // beyond Cout multiply zero-filled LDS columns and are never stored.
                auto load_b = [&](int tap) {
                    const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * SX + colx;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(x1 + toff));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(x2 + toff));
                    return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                bf16x8 af[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int colb = ((nh * 4 + t) * 16 + 4 * (r & 3)) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(y1 + colb));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(y2 + colb));
                    af[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                bf16x8 bcur = load_b(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    bf16x8 bnext = bcur;
                    if (tap < 8) bnext = load_b(tap + 1);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bcur, acc[tap][t], 0, 0, 0);
                    bcur = bnext;
                }
                // issue order: [8 A reads + 2 B reads] then 8 x { 2 B reads of the next tap, 4 MFMAs of this tap }, 4 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
                for (int tap = 0; tap < 8; ++tap) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        } else {
#pragma unroll 1
            for (int ks = 0; ks < KROWS / 4; ++ks) {
                const int k = ks * 4 + q;
                float af[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    af[t] = *reinterpret_cast<const float*>(ytile + k * SY + ((nh * 4 + t) * 16 + r) * 4);
                const int ik = xsq(k < KA_BOARD ? k : 0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int toff = (tap / 3 - 1) * PW + (tap % 3 - 1);
                    const float bv = *reinterpret_cast<const float*>(xtile + (ik + toff) * SX + (cq * 16 + r) * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bv, acc[tap][t], 0, 0, 0);
                }
            }
        }
        }
        if (NBUF >= 2) __syncthreads();
    };
    if (NSETS == 2) {
        for (int b = bbeg; b < bend; b += 2) {           // board b+1 is odd within the range -> set 1, b+2 even -> set 0
            board_iter(b, ry[NSETS - 1], rx[NSETS - 1]);
            if (b + 1 < bend) board_iter(b + 1, ry[0], rx[0]);
        }
    } else {
#pragma unroll 1
        for (int b = bbeg; b < bend; ++b) board_iter(b, ry[0], rx[0]);
    }

    // partial slab: [split][tap][n][c], c contiguous (16 lanes -> 64 B runs)
    if (c_ok) {
        const int c = c0 + cq * 16 + r;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + (nh * 4 + t) * 16 + q * 4 + i;
                    if (n < a.Cout && c < a.Cin)
                        a.slab[(((size_t)split * 9 + tap) * a.Cout + n) * a.Cin + c] = acc[tap][t][i];
                }
    }
}

// ---------------------------------------------------------------- flat-K form, software-pipelined (bf16, 128 x 64 x 9 slab)
// The same slab, tiles, operand reads and summation as wgrad_kernel<bf16_t, 128>, restructured around what its counters
// showed (profiles/r03_mfma_sq_counters.json: matrix pipe 53 % busy, 41 % of the wave cycles issue stalls): every k-step
// began with ~45 dependent address instructions followed by twelve LDS reads whose latency nothing covered, and both
// waves of a SIMD restart in phase after the per-board barrier.  Here
//   * a k-step belongs to the board its FIRST row lies in and may run on into the next board, whose tile was published one
//     barrier earlier (the staging runs two boards ahead of the multiplication instead of one; still a ring of three
//     tiles: board j's slot is rewritten with board j+3 only after the barrier that ends iteration j).  Every k-step
//     therefore reads published tiles only -- also the first one of the next iteration --
//   * so the operand fragments of step s+1 (four dY fragments, the first X fragment) and its row addresses are fetched
//     under the MFMAs of step s, across board boundaries and barriers alike: no step starts with an LDS round trip;
//   * row addresses come from one wave-uniform offset (first row of the step within its board) and two per-lane
//     constants: a compare, a select, p/9 as (57 p) >> 9 and two multiply-adds per row instead of the divisions /
//     conditional pointer chains of the first form (45 -> ~20 vector instructions per step);
//   * the per-channel coefficients of the fused input transform are re-read per board (L2) instead of living in 24
//     registers across the MFMA loop, which pays for the second fragment set.
// Bit-identical slabs (same products, same order within a lane's accumulator: steps in flat-row order).
struct RowPtr { uint32_t y1, y2, x1, x2; };

// NW = 8: 512 threads, 128 x 64 x 9 slab per workgroup (the CU is full: 2 waves x ~240 registers per SIMD, 160 KB of LDS).
// NW = 4 ("lite"): 256 threads -- ONE wave per SIMD, which the in-step prefetch keeps busy without a partner -- and a
//   128 x 32 x 9 slab: 256 registers per SIMD lane and 125 KB of LDS, so that a 512-thread board kernel workgroup (<= 80
//   VGPRs, <= 32 KB: tail_bwd_fused, block_dx) fits on the SAME CU beside it.  The HBM-bound kernels of the backward then run
//   under the weight gradients on all 256 CUs instead of sharing the chip by CU partition.
// NW = 8, TC = 32 ("half"): 512 threads on a 128 x 32 x 9 slab; the two waves of a SIMD split the NINE TAPS (5 + 4) of the same
//   64 x 16 output tile instead of owning different tiles: 80 accumulator registers per wave, <= 168 registers in all, so
//   that two waves per SIMD (336 registers) leave room for a board kernel workgroup (2 x 80) -- the co-resident form.
template <bool FUSED, int NW, int TC>
__device__ __forceinline__ void wgrad_flat_body(const WgradArgs& a) {
    typedef Elem<bf16_t> E;
    typedef bf16x8 vec16;
    constexpr int TN = 128, NTHR = 64 * NW, P16 = 8;
    constexpr int CT = TC / 16, NGRP = NW / (2 * CT), NTAP = NGRP == 1 ? 9 : 5;       // c-tiles, tap groups, taps per wave (at most)
    static_assert(NGRP == 1 || NGRP == 2, "wgrad_flat_kernel: waves = 2 x c-tiles x {1, 2} tap groups");
    // lite: the haloed X image is 10 squares wide (the right halo of a board row IS the left halo of the next) instead of 17:
    // 103 KB of LDS instead of 125 -- measured with a synthetic MFMA kernel (tools/_diag/coresidency.py): a board kernel
    // workgroup shares the CU beside 96 KB, not beside 124 KB -- at the price of one 2-way bank conflict in the transpose
    // reads that straddle a board-row wrap (the 17-wide image has none: a wrap advances the square index by 9 = 1 mod 8)
    constexpr int SY = WG<bf16_t, TN>::SY, SX = TC * 2 + 32, KROWS = WG<bf16_t, TN>::KROWS, PW = TC == 32 ? 10 : 17;
    constexpr int XSQ = 11 * PW + 1, YB = KROWS * SY, TILE = (YB + XSQ * SX + 15) / 16 * 16;
    constexpr int PY = TN * 2 / 16, PX = TC * 2 / 16;
    constexpr int NY = (KA_BOARD * PY + NTHR - 1) / NTHR, NX = (KA_BOARD * PX + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nh = wave & 1, cq = (wave >> 1) % CT, th = wave / (2 * CT);      // n half, c-tile, tap group (SIMD partners: w, w + 4)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;         // XCD-aware (tile, split) map: see wgrad_kernel
    const int tile = slot % a.ntiles;
    const int split = xcd + 8 * (slot / a.ntiles);
    if (split >= a.nsplit) return;
    const int tn = tile % a.ntn, tc = tile / a.ntn;
    const int n0 = tn * TN, c0 = tc * TC;
    const int bbeg = split * a.boards_per_split;
    const int bend = min(a.B, bbeg + a.boards_per_split);
    const int nb = max(0, bend - bbeg);

    f32x4 acc[NTAP][4];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 3 * TILE / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};

    // ---- staging (all eight waves; one register set, two boards ahead of its LDS write)
    const int yj = tid % PY, xj = tid % PX;
    const bool ycol_ok = n0 + yj * P16 < a.Cout, xcol_ok = c0 + xj * P16 < a.Cin;
    const bool has_aff = FUSED && a.in_scale != nullptr;
    vec16 ry[NY], rx[NX];
    auto load_board = [&](int b) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int row = (tid + i * NTHR) / PY;
            ry[i] = (row < KA_BOARD && ycol_ok)
                        ? *reinterpret_cast<const vec16*>(static_cast<const char*>(a.dy) + ((size_t)(b * KA_BOARD + row) * a.Cout + n0 + yj * P16) * 2)
                        : vec16{};
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = (tid + i * NTHR) / PX;
            rx[i] = (row < KA_BOARD && xcol_ok)
                        ? *reinterpret_cast<const vec16*>(static_cast<const char*>(a.x) + ((size_t)(b * KA_BOARD + row) * a.Cin + c0 + xj * P16) * 2)
                        : vec16{};
        }
    };
    auto store_board = [&](int b, uint32_t tbase) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int row = (tid + i * NTHR) / PY;
            if (row < KA_BOARD) *reinterpret_cast<vec16*>(smem + tbase + row * SY + yj * 16) = ry[i];
        }
        const bool xform = FUSED && xcol_ok && (has_aff || a.relu || a.in_bias);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // scale | shift | per-board bias of four of this thread's 8 channels, re-read per board (L2) and four at a time:
            // 12 registers alive during the transform instead of 24 across the whole MFMA loop
            f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f}, sh4 = f32x4{0.f, 0.f, 0.f, 0.f}, bi4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (xform) {
                const int cc = c0 + xj * P16 + 4 * h;
                if (has_aff) { sc4 = *reinterpret_cast<const f32x4*>(a.in_scale + cc); sh4 = *reinterpret_cast<const f32x4*>(a.in_shift + cc); }
                if (a.in_bias) bi4 = *reinterpret_cast<const f32x4*>(a.in_bias + (size_t)b * a.Cin + cc);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int row = (tid + i * NTHR) / PX;
                if (row >= KA_BOARD) continue;
                if (xform) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float f = (float)rx[i][4 * h + e];
                        if (has_aff) f = fmaf(f, sc4[e], sh4[e]);
                        if (a.relu) f = fmaxf(f, 0.f);
                        if (a.in_bias) f += bi4[e];
                        rx[i][4 * h + e] = (__bf16)f;
                    }
                }
                if (h == 1) {
                    const int sq = (row / 9 + 1) * PW + (row % 9) + 1;
                    *reinterpret_cast<vec16*>(smem + tbase + YB + sq * SX + xj * 16) = rx[i];
                }
            }
        }
    };

    // ---- MFMA side
    const int ntn_valid = min(4, max(0, (a.Cout - n0 - nh * 64 + 15) / 16));
    const bool active = (c0 + cq * 16 < a.Cin) && ntn_valid > 0;       // wave-uniform
    // k-slot (q, j) <-> row 4q + j (j < 4) / 16 + 4q + (j - 4) of the step, for both operands (see wgrad_kernel)
    const int l1 = 4 * q + (r >> 2);
    const uint32_t colA = (uint32_t)((nh * 64 + 4 * (r & 3)) * 2);
    const uint32_t colX = (uint32_t)(YB + (PW + 1) * SX + (cq * 16 + 4 * (r & 3)) * 2);  // (+PW+1 squares: image index of square 0)
    // rows p (< 81: in the tile at sbA) / p - 81 (in the tile at sbB) of a step whose first row is Fj within its board; `edge`:
    // the board after sbA's lies outside the range -- its rows read the zero row 81 of sbA's dY tile (the product vanishes)
    auto ptrs = [&](int Fj, uint32_t sbA, uint32_t sbB, bool edge) {
        RowPtr o;
        int p1 = l1 + Fj, p2 = p1 + 16;
        const bool t1 = p1 >= KA_BOARD, t2 = p2 >= KA_BOARD;
        p1 = t1 ? p1 - KA_BOARD : p1; p2 = t2 ? p2 - KA_BOARD : p2;
        uint32_t b1 = t1 ? sbB : sbA, b2 = t2 ? sbB : sbA;
        int y1r = p1, y2r = p2;
        if (edge) { y1r = t1 ? KA_BOARD : p1; y2r = t2 ? KA_BOARD : p2; b1 = sbA; b2 = sbA; }
        const int d1 = (p1 * 57) >> 9, d2 = (p2 * 57) >> 9;               // p / 9 for 0 <= p < 81
        o.y1 = b1 + colA + (uint32_t)(y1r * SY); o.y2 = b2 + colA + (uint32_t)(y2r * SY);
        o.x1 = b1 + colX + (uint32_t)((p1 + (PW - 9) * d1) * SX); o.x2 = b2 + colX + (uint32_t)((p2 + (PW - 9) * d2) * SX);
        return o;
    };
    auto rd = [&](uint32_t off) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + off)); };
    auto load_a = [&](const RowPtr& pt, int t) {
        bf16x4 lo = rd(pt.y1 + t * 32), hi = rd(pt.y2 + t * 32);
        return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_b = [&](const RowPtr& pt, int tap) {
        const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * SX;
        bf16x4 lo = rd(pt.x1 + toff), hi = rd(pt.x2 + toff);
        return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // One k-step of a wave that owns taps T0 .. T0 + NT - 1 (NT x 4 MFMAs).  On entry A[0..3] hold its four dY fragments and
    // Bs[0], Bs[1] the X fragments of its first two taps -- all fetched under the MFMAs of the step before.  X fragments run
    // two taps ahead through five register slots; the last two taps go n-tile by n-tile, so that each dY fragment register is
    // free a few MFMAs before the step ends and is refilled IN PLACE with the following step's fragment (no second fragment
    // set: the slab leaves no room for one).
#define KA_MM(tap_, t_, slot_) acc[(tap_) - T0][t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t_], Bs[slot_], acc[(tap_) - T0][t_], 0, 0, 0)
#define KA_WG_GRP(d_, m_) __builtin_amdgcn_sched_group_barrier(0x100, d_, 0); __builtin_amdgcn_sched_group_barrier(0x008, m_, 0)
    auto kstep = [&](auto t0c, auto ntc, const RowPtr& pc, const RowPtr& pn, bf16x8 (&A)[4], bf16x8 (&Bs)[5]) __attribute__((always_inline)) {
        constexpr int T0 = decltype(t0c)::value, NT = decltype(ntc)::value, LA = NT - 2, LB = NT - 1;   // the two tail taps
#pragma unroll
        for (int i = 0; i < NT - 2; ++i) {
            Bs[(i + 2) % 5] = load_b(pc, T0 + i + 2);
#pragma unroll
            for (int t = 0; t < 4; ++t) KA_MM(T0 + i, t, i % 5);
        }
        Bs[0] = load_b(pn, T0); KA_MM(T0 + LA, 0, LA % 5); KA_MM(T0 + LA, 1, LA % 5); KA_MM(T0 + LB, 0, LB % 5);
        Bs[1] = load_b(pn, T0 + 1); A[0] = load_a(pn, 0); KA_MM(T0 + LB, 1, LB % 5);
        A[1] = load_a(pn, 1); KA_MM(T0 + LA, 2, LA % 5); KA_MM(T0 + LB, 2, LB % 5);
        A[2] = load_a(pn, 2); KA_MM(T0 + LA, 3, LA % 5); KA_MM(T0 + LB, 3, LB % 5);
        A[3] = load_a(pn, 3);
        // issue order pinned: {LDS reads, MFMAs} groups exactly as written
#pragma unroll
        for (int i = 0; i < NT - 2; ++i) { KA_WG_GRP(2, 4); }
        KA_WG_GRP(2, 3); KA_WG_GRP(4, 1); KA_WG_GRP(2, 2); KA_WG_GRP(2, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
    typedef std::integral_constant<int, 9> I9;
    const int tap0 = (NGRP == 2 && th == 1) ? 5 : 0;              // first tap of this wave (wave-uniform)

    // ---- prologue: tiles 0 and 1 published, board 2 in registers
    __syncthreads();                                    // zero fill complete
    if (nb > 0) { load_board(bbeg); store_board(bbeg, 0); }
    if (nb > 1) { load_board(bbeg + 1); store_board(bbeg + 1, TILE); }
    if (nb > 2) load_board(bbeg + 2);
    __syncthreads();

    uint32_t sA = 0, sB = TILE, sC = 2 * TILE;          // slots of boards j, j + 1, j + 2
    int Fj = 0, j = 0;                                  // first row of the next step within board j
    RowPtr pc = ptrs(0, sA, sB, nb <= 1);
    bf16x8 A[4], Bs[5];
#pragma unroll
    for (int t = 0; t < 4; ++t) A[t] = load_a(pc, t);
    Bs[0] = load_b(pc, tap0); Bs[1] = load_b(pc, tap0 + 1); Bs[2] = Bs[3] = Bs[4] = bf16x8{};
    if (nb > 2) {                                       // iteration 0's staging: tile 2, board 3 into the registers
        store_board(bbeg + 2, sC);
        if (nb > 3) load_board(bbeg + 3);
    }
    // (inactive waves -- tiles beyond Cout / Cin of a small layer -- multiply zero-filled LDS columns: no run-time guard
    //  around the MFMA stream, which would send the accumulators through scratch)
    (void)ntn_valid; (void)active;
    const int S = (KA_BOARD * nb + 31) / 32;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const int Fn = Fj + 32;
        const bool adv = Fn >= KA_BOARD;                // the following step opens board j + 1
        const RowPtr pn = ptrs(adv ? Fn - KA_BOARD : Fn, adv ? sB : sA, adv ? sC : sB, (adv ? j + 2 : j + 1) >= nb);
        if constexpr (NGRP == 1) kstep(I0{}, I9{}, pc, pn, A, Bs);
        else {
            if (th == 0) kstep(I0{}, I5{}, pc, pn, A, Bs);
            else kstep(I5{}, I4{}, pc, pn, A, Bs);
        }
        pc = pn;
        Fj = adv ? Fn - KA_BOARD : Fn;
        if (adv) {
            const uint32_t s0 = sA; sA = sB; sB = sC; sC = s0;
            KA_LDS_BARRIER();                           // tile j + 2 is published; tile j may be rewritten
            ++j;
            if (j + 2 < nb) {
                store_board(bbeg + j + 2, sC);
                if (j + 3 < nb) load_board(bbeg + j + 3);
            }
        }
    }

#undef KA_MM
#undef KA_WG_GRP
    // partial slab: [split][tap][n][c], c contiguous (16 lanes -> 64 B runs)
    if (c0 + cq * 16 < a.Cin) {
        const int c = c0 + cq * 16 + r;
        const int ntap = NGRP == 1 ? 9 : (th == 0 ? 5 : 4);
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + (nh * 4 + t) * 16 + q * 4 + i;
                    if (k < ntap && n < a.Cout && c < a.Cin)
                        a.slab[(((size_t)split * 9 + tap0 + k) * a.Cout + n) * a.Cin + c] = acc[k][t][i];
                }
    }
}

template <bool FUSED, int NW, int TC>
__global__ __launch_bounds__(64 * NW, 1) void wgrad_flat_kernel(WgradArgs a) { wgrad_flat_body<FUSED, NW, TC>(a); }
// the co-resident form: 176 registers (2 x 176 + 2 x 80 = the 512 of a SIMD lane); the attribute takes a literal only
__global__ __launch_bounds__(512, 3) void wgrad_half_plain_kernel(WgradArgs a) { wgrad_flat_body<false, 8, 32>(a); }
__global__ __launch_bounds__(512, 3) void wgrad_half_fused_kernel(WgradArgs a) { wgrad_flat_body<true, 8, 32>(a); }

// dW[n][c][tap] (Cout, Cin_real, 3, 3) = sum_s slab[s][tap][n][c]; optional accumulate into dW.
// One thread sums 4 consecutive c of one (tap, n) over the splits: the slab reads -- nsplit times the bytes of the
// result -- are whole coalesced 16-byte pieces in slab order; only the 4-byte result stores are strided (by 9).
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int Cout,
                                    int Cin, int Cin_real, int accumulate) {
    const int c4n = Cin >> 2;                                      // Cin % 4 == 0 (checked by the launcher)
    const size_t total = (size_t)9 * Cout * c4n, sstride = (size_t)9 * Cout * c4n;   // in 16-byte pieces
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(slab);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t tn = i / c4n;
        const int n = (int)(tn % Cout), tap = (int)(tn / Cout);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < nsplit; ++k) acc += s4[i + k * sstride];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 * 4 + e;
            if (c < Cin_real) {
                const size_t o = ((size_t)n * Cin_real + c) * 9 + tap;
                dw[o] = accumulate ? dw[o] + acc[e] : acc[e];
            }
        }
    }
}

}  // namespace

// n-tile width / workgroup size of the kernel.  Alone on the chip the 64-wide form (256 threads, 2 workgroups per CU) is
// 16 % faster for a plain input (0.40 vs 0.47 ms at B=4096, C=256) and slower with the fused BatchNorm+ReLU+bias
// input transform, which it repeats for twice as many n-tiles.  Inside the training step, where the kernel shares the
// chip with the main stream, the two forms are within 0.3 % of each other (A/B on one box), so the 128-wide form stays
// the default; KA_WGRAD_TN=64 selects the other.  Both give the same split count.
static int wgrad_tn(bool fused_input) {
    (void)fused_input;
    if (const char* e = getenv("KA_WGRAD_TN")) { if (atoi(e) == 64) return 64; }
    return 128;
}

// Which kernel takes a bf16 launch: 1 = wgrad_kernel (128 x 64 slab, 8 waves), 2 = the software-pipelined flat-K form with the
// same slab, 3 = its 4-wave "lite" form (128 x 32 slab; leaves room for a co-resident board kernel workgroup).  KA_WGRAD_V.
static int wgrad_variant(int dtype) {
    if (dtype != KA_DTYPE_BF16 || wgrad_tn(false) != 128) return 1;
    if (const char* e = getenv("KA_WGRAD_V")) { const int v = atoi(e); if (v >= 1 && v <= 4) return v; }
    return 1;
}

// target_wgs: CUs to aim for (0 = all 256).  Fewer leaves CUs free for kernels that run concurrently on another
// stream (the engine overlaps wgrad with the HBM-bound backward kernels and asks for 192).
static int wgrad_splits_for(int B, int Cin, int Cout, int target_wgs, int tc) {
    const int tn = 128;            // the 64-wide variant has twice the tiles and twice the workgroups per CU: same count
    const int tiles = ((Cout + tn - 1) / tn) * ((Cin + tc - 1) / tc);
    if (const char* e = getenv("KA_WGRAD_WGS")) { const int v = atoi(e); if (v > 0) target_wgs = v; }   // experiments
    int s = (target_wgs > 0 ? target_wgs : 256) / tiles;
    if (s < 1) s = 1;
    if (s > B) s = B;
    const int bps = (B + s - 1) / s;
    return (B + bps - 1) / bps;
}

// number of partial slabs the caller must provide room for ([splits][9][Cout][Cin] floats): the largest count any kernel
// variant would use for this shape (the launch picks its own, never more)
extern "C" int ka_wgrad_splits(int B, int Cin, int Cout, int target_wgs) {
    const int a = wgrad_splits_for(B, Cin, Cout, target_wgs, kTC), b = wgrad_splits_for(B, Cin, Cout, 0, 32);
    return a > b ? a : b;
}

template <typename T, int TN, bool FUSED>
static int launch_wgrad(const WgradArgs& a, dim3 grid, hipStream_t st) {
    constexpr int NBUF = (sizeof(T) == 2 && TN == 128) ? 3 : (sizeof(T) == 2 ? 2 : 1);   // as in the kernel
    const size_t lds = (size_t)NBUF * (WG<T, TN>::KROWS * WG<T, TN>::SY + 11 * WG<T, TN>::PW * WG<T, TN>::SX);
    static std::atomic<unsigned long long> attr_done{0};   // per instantiation: devices already configured
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_kernel<T, TN, FUSED>), attr_done, "wgrad")) return rc;
    hipLaunchKernelGGL((wgrad_kernel<T, TN, FUSED>), grid, dim3(TN * 4), lds, st, a);
    return ka_check_launch("wgrad");
}

extern "C" int ka_conv3x3_wgrad(const void* dy, const void* x, const float* in_scale, const float* in_shift,
                                const float* in_bias, int relu, float* slab, float* dw, int B, int Cin, int Cin_real,
                                int Cout, int accumulate, int target_wgs, int dtype, void* stream) {
    KA_REQUIRE(dy && x && slab && dw && B > 0, "wgrad: null tensor");
    KA_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0 && Cin_real <= Cin, "wgrad: need Cin,Cout %% 16 == 0");
    KA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wgrad: scale/shift must come together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tn = wgrad_tn(in_scale || in_bias || relu);
    const int variant = wgrad_variant(dtype);
    const int tcw = variant >= 3 ? 32 : kTC;
    // (the lite form always spreads over all CUs: board kernels share its CUs instead of taking the ones it leaves free)
    const int nsplit = wgrad_splits_for(B, Cin, Cout, variant == 3 ? 0 : target_wgs, tcw);    // (4: the caller's CU target)
    const int bps = (B + nsplit - 1) / nsplit;
    const int ntn = (Cout + tn - 1) / tn, ntiles = ntn * ((Cin + tcw - 1) / tcw);
    WgradArgs a{dy, x, in_scale, in_shift, in_bias, slab, B, Cin, Cout, relu, bps, ntn, ntiles, nsplit};
    dim3 grid(8 * ntiles * ((nsplit + 7) / 8));
    int rc;
    const bool fused = in_scale || in_bias || relu;
#define KA_WG(T_, TN_) (fused ? launch_wgrad<T_, TN_, true>(a, grid, st) : launch_wgrad<T_, TN_, false>(a, grid, st))
#define KA_WGF(NW_, TC_)                                                                                                    \
    {                                                                                                                       \
        const size_t lds = 3 * (((size_t)WG<bf16_t, 128>::KROWS * WG<bf16_t, 128>::SY + (11 * (TC_ == 32 ? 10 : 17) + 1) * (2 * TC_ + 32) + 15) / 16 * 16); \
        static std::atomic<unsigned long long> d0{0}, d1{0};                                                                \
        if (fused) {                                                                                                        \
            if (int r2 = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_flat_kernel<true, NW_, TC_>), d1, "wgrad (flat)")) return r2; \
            hipLaunchKernelGGL((wgrad_flat_kernel<true, NW_, TC_>), grid, dim3(64 * NW_), lds, st, a);                      \
        } else {                                                                                                            \
            if (int r2 = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_flat_kernel<false, NW_, TC_>), d0, "wgrad (flat)")) return r2; \
            hipLaunchKernelGGL((wgrad_flat_kernel<false, NW_, TC_>), grid, dim3(64 * NW_), lds, st, a);                     \
        }                                                                                                                   \
        rc = ka_check_launch("wgrad (flat)");                                                                               \
    }
    if (variant == 2) KA_WGF(8, 64)
    else if (variant == 3) KA_WGF(4, 32)
    else if (variant == 4) {
        const size_t lds = 3 * (((size_t)WG<bf16_t, 128>::KROWS * WG<bf16_t, 128>::SY + (11 * 10 + 1) * (2 * 32 + 32) + 15) / 16 * 16);
        static std::atomic<unsigned long long> d0{0}, d1{0};
        if (fused) {
            if (int r2 = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_half_fused_kernel), d1, "wgrad (half)")) return r2;
            hipLaunchKernelGGL(wgrad_half_fused_kernel, grid, dim3(512), lds, st, a);
        } else {
            if (int r2 = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_half_plain_kernel), d0, "wgrad (half)")) return r2;
            hipLaunchKernelGGL(wgrad_half_plain_kernel, grid, dim3(512), lds, st, a);
        }
        rc = ka_check_launch("wgrad (half)");
    }
    else if (dtype == KA_DTYPE_BF16) rc = tn == 64 ? KA_WG(bf16_t, 64) : KA_WG(bf16_t, 128);
    else if (dtype == KA_DTYPE_F32) rc = tn == 64 ? KA_WG(float, 64) : KA_WG(float, 128);
#undef KA_WG
#undef KA_WGF
    else { ka_set_error("wgrad: unknown dtype %d", dtype); return KA_ERR_ARG; }
    if (rc) return rc;
    const size_t total = (size_t)9 * Cout * (Cin / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, dw, nsplit, Cout, Cin, Cin_real,
                       accumulate);
    return ka_check_launch("wgrad_reduce");
}
