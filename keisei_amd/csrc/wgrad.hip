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
    int stagger;             // waves 4-7 stage the next board AFTER their MFMAs instead of before them (see board_iter)
    unsigned long long* stamps;   // diagnostic only (ka_debug_conv_stamps): [workgroup][8]; null in production
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
    const bool late = NBUF >= 2 && TN == 128 && a.stagger && __builtin_amdgcn_readfirstlane(wave) >= 4;
    // one board; ys/xs = the register set of board b+1 (compile-time choice: the loop below is unrolled by two)
    auto board_iter = [&](int b, vec16 (&ys)[NY], vec16 (&xs)[NX]) {
        if (NBUF == 1) {
            __syncthreads();                 // previous board fully consumed
            store_board(b, ys, xs);
            __syncthreads();
            if (b + 1 < bend) load_board(b + 1, ys, xs);      // in flight during the MFMA phase
        } else {
            // The two waves of a SIMD (w and w + 4) run the same program between the same barriers: in lockstep they both
            // stage (LDS writes, the input transform, the next loads) and then both queue for the matrix pipe.  With `late`
            // waves 4-7 multiply first and stage afterwards, so one partner's staging sits beside the other's MFMAs
            // (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Either order is legal inside one barrier interval: the
            // tile written (board b+1) is read from the next iteration on, the tiles read (boards b, b-1) were complete at
            // the last barrier.
            const int cur = (b - bbeg) % NBUF;
            if (!late) stage_next(b, ys, xs);
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
        if (NBUF >= 2) {
            if (late) stage_next(b, ys, xs);
            __syncthreads();
        }
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

// ---------------------------------------------------------------- bf16, 128-wide tile: the lean form
// wgrad_kernel<bf16_t, 128> issues 2.7 vector instructions per MFMA (profiles/r03_mfma_sq_counters.json): with the matrix
// instruction holding the SIMD's vector issue for half of its 16 cycles, two such waves per SIMD are issue-bound at a matrix pipe
// 53 % busy.  Where they come from: every k-step opens with a dependent chain of ~40 of them (flat row -> board / square ->
// haloed tile index, two divisions by 9) in front of its first LDS read, every board re-derives its staging addresses
// (~200), and the slab store branches per element.  Same tiles, ring, k-slot permutation, MFMA order and split-K map here
// (bit-identical slabs), but
//   * the flat row of a k-step is walked by the SCALAR unit (board offset and first row of the step), a lane adds its row
//     slot, wraps once, and takes the haloed-tile offset of its square from an 82-entry LDS table;
//   * the row offsets of step s+1 are computed behind the reads of step s (they do not depend on staged data), so a step
//     starts with its LDS reads;
//   * staging addresses are lane constants; the slab leaves without per-element branches when the tile is whole.
template <bool FUSED>
__global__ __launch_bounds__(512, 1) void wgrad_flat_kernel(WgradArgs a) {
    constexpr int NTHR = 512, SY = 128 * 2 + 32, SX = kTC * 2 + 32, KROWS = 82, PW = 17, XSQ = 11 * PW;
    constexpr int YB = KROWS * SY, TILE = YB + XSQ * SX, XTAB = 3 * TILE;      // [3 tiles][dY rows | haloed X squares] | square table
    constexpr int kTapBias = (PW + 1) * SX;
    // PIPE: the A fragments of the next k-step of a board requested under this step's MFMAs (a second register set): built,
    // bit-identical, and worth nothing (plain input 327 vs 326 us, profiles/NOTES_r04.md) -- off
    constexpr bool PIPE = false;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (provably wave-uniform: the k-step walk below stays on the scalar unit)
    const int r = lane & 15, q = lane >> 4;
    const int nh = wave & 1, cq = wave >> 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;       // (the XCD-aware tile / split map of wgrad_kernel)
    const int tile = slot % a.ntiles;
    const int split = xcd + 8 * (slot / a.ntiles);
    if (split >= a.nsplit) return;
    const int tn = tile % a.ntn, tc = tile / a.ntn;
    const int n0 = tn * 128, c0 = tc * kTC;
    const int bbeg = split * a.boards_per_split;
    const int bend = min(a.B, bbeg + a.boards_per_split), nb = bend - bbeg;

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 3 * TILE / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
    if (tid < KROWS) {                                       // byte offset of square p in the haloed X tile; entry 81: a square of the halo-free
        const int p = tid < KA_BOARD ? tid : 0;              // board whose dY row is the zero pad row
        reinterpret_cast<int*>(smem + XTAB)[tid] = YB + ((p / 9 + 1) * PW + (p % 9) + 1) * SX;
    }

    // ---- staging roles: lane constants.  The staging shares its SIMD with the other wave's MFMAs, which leave the vector unit
    // one issue slot in two: every vector instruction, exec-mask branch and address add of the staging is time the board period
    // grows by (stamps, tools/_diag/wgrad_tl.py: 900-1100 cycles per board for what used to be ~85 instructions in ten branchy
    // blocks, 500-700 for the 15 below).  So the tensors are read through buffer descriptors -- lane offset fixed for the whole
    // launch, the board a scalar offset, lanes without a piece (rows past the board, columns past the tensor) parked out of range,
    // where a load returns zeros -- and the LDS stores are unconditional: a lane without a row writes into the pad bytes of
    // row 0 / square (1, 1), which nothing reads.
    const int yj = tid & 15, xj = tid & 7;
    const bool ycol_ok = n0 + yj * 8 < a.Cout, xcol_ok = c0 + xj * 8 < a.Cin;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (unsigned)a.B * (KA_BOARD * 2u) * (unsigned)a.Cout, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (unsigned)a.B * (KA_BOARD * 2u) * (unsigned)a.Cin, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FUSED ? a.in_bias : nullptr), 0,
                                                                          FUSED && a.in_bias ? (unsigned)a.B * 4u * (unsigned)a.Cin : 0u, 0x00020000);
    constexpr int kPark = 0x7fffffff;
    int ldsY[3], gY[3], ldsX[2], gX[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (tid >> 4) + 32 * i;
        gY[i] = row < KA_BOARD && ycol_ok ? (row * a.Cout + n0 + yj * 8) * 2 : kPark;
        ldsY[i] = row < KA_BOARD ? row * SY + yj * 16 : 256;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 64 * i;
        gX[i] = row < KA_BOARD && xcol_ok ? (row * a.Cin + c0 + xj * 8) * 2 : kPark;
        ldsX[i] = row < KA_BOARD ? YB + ((row / 9 + 1) * PW + (row % 9) + 1) * SX + xj * 16 : YB + (PW + 1) * SX + kTC * 2;
    }
    const int gB = xcol_ok ? (c0 + xj * 8) * 4 : kPark;
    // (columns past the tensor: scale = shift = 0 and a bias that reads as zeros keep their squares at exactly zero through the transform)
    float sc[FUSED ? 8 : 1] = {}, sh[FUSED ? 8 : 1] = {};
    f32x4 rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = rb0;
    const bool has_aff = FUSED && a.in_scale != nullptr;
    if (has_aff && xcol_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale[c0 + xj * 8 + e]; sh[e] = a.in_shift[c0 + xj * 8 + e]; }
    }
    // the last round of either tensor has rows for the first waves only (dY: rows 64 + 4 wave .., X: rows 64 + 8 wave ..): the
    // others skip it, wave-uniformly (for X that is the whole second transform)
    const bool y2 = 64 + 4 * wave < KA_BOARD, x1 = 64 + 8 * wave < KA_BOARD;
    bf16x8 ry[3], rx[2];
    auto load_board = [&](int b) {
        const int sy = __builtin_amdgcn_readfirstlane(b * (KA_BOARD * 2) * a.Cout), sx = __builtin_amdgcn_readfirstlane(b * (KA_BOARD * 2) * a.Cin);
        if (FUSED && a.in_bias) {
            const int sb = __builtin_amdgcn_readfirstlane(b * 4 * a.Cin);
            rb0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, gB, sb, 0));
            rb1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, gB + 16, sb, 0));
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < 2 || y2) ry[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_y, gY[i], sy, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i < 1 || x1) rx[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, gX[i], sx, 0));
    };
    auto store_board = [&](int toff) {                       // into the tile at byte offset toff of the ring
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < 2 || y2) *reinterpret_cast<bf16x8*>(smem + toff + ldsY[i]) = ry[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !x1) continue;
            bf16x8 v = rx[i];
            if (FUSED && (has_aff || a.relu || a.in_bias)) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
                if (has_aff) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                }
                if (a.in_bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f[e] += rb0[e]; f[4 + e] += rb1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
            }
            *reinterpret_cast<bf16x8*>(smem + toff + ldsX[i]) = v;
        }
    };

    const int ntn_valid = min(4, max(0, (a.Cout - n0 - nh * 64 + 15) / 16));
    const bool skip = !(c0 + cq * 16 < a.Cin) || ntn_valid == 0;      // wave-uniform
    const bool late = a.stagger && __builtin_amdgcn_readfirstlane(wave) >= 4;

    // ---- the flat row walk.  Scalar: kb = first flat row of the next k-step, as (byte offset of its board's tile in the ring,
    // row inside that board).  Lane: its two rows of the step are kb + rsub and kb + rsub + 16 (the k-slot permutation of
    // wgrad_kernel), each in this board or -- past row 80 -- at the start of the next one; rows past the range read the zero
    // pad row 81 of a tile (only the last step has them).
    const int rsub = 4 * q + (r >> 2), kend = KA_BOARD * nb;
    int s_toff = 0, s_p = 0, s_k = 0;                         // scalar state of the NEXT step to prepare
    int yo[2], xo[2];                                         // LDS byte offsets of the prepared step: dY rows, X squares
    const int coly = (nh * 64 + 4 * (r & 3)) * 2, colx = (cq * 16 + 4 * (r & 3)) * 2;
    auto prepare = [&]() {                                    // offsets of step s_k / 32, then advance the scalar state by one step
        const int toff_next = s_toff + TILE == 3 * TILE ? 0 : s_toff + TILE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int p = s_p + rsub + 16 * h;
            const bool wrap = p >= KA_BOARD;
            p = wrap ? p - KA_BOARD : p;
            int tf = wrap ? toff_next : s_toff;
            if (s_k + 32 > kend) {                            // (uniform: the last step of the range)
                const bool pad = s_k + rsub + 16 * h >= kend;
                p = pad ? KA_BOARD : p;
                tf = pad ? s_toff : tf;
            }
            yo[h] = tf + p * SY + coly;
            xo[h] = tf + *reinterpret_cast<const int*>(smem + XTAB + p * 4) + colx - kTapBias;    // (every tap offset then is a non-negative immediate)
        }
        s_k += 32; s_p += 32;
        if (s_p >= KA_BOARD) { s_p -= KA_BOARD; s_toff = toff_next; }
    };

    if (a.stamps && tid == 0) {
        a.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
        a.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    if (nb > 0) load_board(bbeg);
    // (every barrier of this kernel orders LDS traffic only: __syncthreads() would also drain the vector-memory counter, i.e. wait
    //  at each board for the prefetch loads of the board after next -- a full HBM round trip for the waves that issue them last)
    KA_LDS_BARRIER();                                         // zero fill and the square table complete
    if (nb > 0) {
        store_board(0);
        if (nb > 1) load_board(bbeg + 1);
    }
    prepare();                                                // step 0 (reads the table: behind the barrier above)
    KA_LDS_BARRIER();
    int ks = 0;
#ifdef KA_DIAG_WGRAD_TL
    // diagnostic build (tools/_diag/wgrad_tl.py): per-wave phase stamps of workgroup 0, boards 8..23 ->
    // stamps[4096 * 8 + ((board - 8) * 8 + wave) * 8 + phase]
#define KA_TL(ph) do { if (a.stamps && blockIdx.x == 0 && jl >= 8 && jl < 24 && lane == 0) \
        a.stamps[4096 * 8 + ((jl - 8) * 8 + wave) * 8 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KA_TL(ph) do {} while (0)
#endif
#pragma unroll 1
    for (int jl = 0; jl < nb; ++jl) {
        KA_TL(0);
        const int nxt = ((jl + 1) % 3) * TILE;
        auto stage_next = [&]() {
            if (jl + 1 < nb) {
                store_board(nxt);
                if (jl + 2 < nb) load_board(bbeg + jl + 2);
            }
        };
        if (!late) stage_next();
        KA_TL(1);
        const int ks_hi = jl + 1 == nb ? (KA_BOARD * nb + 31) / 32 : (KA_BOARD * (jl + 1)) / 32;
        if (!skip) {
            // one k-step: 36 MFMAs on the A fragments `ac`; PIPE (the plain-input form, which has the registers): the A fragments of the
            // NEXT step of this board are requested into `an` right behind this step's first B reads and land under its MFMAs
            auto load_a = [&](bf16x8 (&f)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + yo[0] + t * 32));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + yo[1] + t * 32));
                    f[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            auto kstep = [&](bf16x8 (&ac)[4], bf16x8 (&an)[4], bool more) {
                const int x0 = xo[0], x1 = xo[1];
                auto load_b = [&](int tap) {
                    const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * SX + kTapBias;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + x0 + toff));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + x1 + toff));
                    return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                bf16x8 bcur = load_b(0), bnx = load_b(1);
                prepare();                                    // the next step's offsets, behind this step's first reads
                if (PIPE && more) load_a(an);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    bf16x8 bn2 = bnx;
                    if (tap < 7) bn2 = load_b(tap + 2);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[t], bcur, acc[tap][t], 0, 0, 0);
                    bcur = bnx; bnx = bn2;
                }
            };
            bf16x8 afA[4];
            if constexpr (PIPE) {
                bf16x8 afB[4];
                if (ks < ks_hi) load_a(afA);
#pragma unroll 1
                for (; ks + 1 < ks_hi; ks += 2) {
                    kstep(afA, afB, true);
                    kstep(afB, afA, ks + 2 < ks_hi);
                }
                if (ks < ks_hi) { kstep(afA, afB, false); ++ks; }
            } else {
#pragma unroll 1
                for (; ks < ks_hi; ++ks) {
                    load_a(afA);
                    kstep(afA, afA, false);
                }
            }
        } else {
            ks = ks_hi;
        }
        KA_TL(2);
        if (late) stage_next();
        KA_TL(3);
        KA_LDS_BARRIER();
        KA_TL(4);
    }
#undef KA_TL

    if (a.stamps && tid == 0) {
        a.stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
        a.stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    }
    // ---- partial slab: [split][tap][n][c], c contiguous (16 lanes -> 64 B runs)
    if (!skip) {
        const int c = c0 + cq * 16 + r;
        float* sl = a.slab + (size_t)split * 9 * a.Cout * a.Cin;
        const bool whole = n0 + nh * 64 + 64 <= a.Cout && c0 + cq * 16 + 16 <= a.Cin;     // wave-uniform
        if (whole) {
            float* base = sl + (size_t)(n0 + nh * 64 + q * 4) * a.Cin + c;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        base[((size_t)tap * a.Cout + t * 16 + i) * a.Cin] = acc[tap][t][i];
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = n0 + (nh * 4 + t) * 16 + q * 4 + i;
                        if (n < a.Cout && c < a.Cin) sl[((size_t)tap * a.Cout + n) * a.Cin + c] = acc[tap][t][i];
                    }
        }
    }
}

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
    if (ka_opt(KA_OPT_WGRAD_TN, 128) == 64) return 64;
    return 128;
}

// target_wgs: CUs to aim for (0 = all 256).  Fewer leaves CUs free for kernels that run concurrently on another
// stream (the engine overlaps wgrad with the HBM-bound backward kernels and asks for 192).
static int wgrad_splits_for(int B, int Cin, int Cout, int target_wgs) {
    const int tc = kTC;
    const int tn = 128;            // the 64-wide variant has twice the tiles and twice the workgroups per CU: same count
    const int tiles = ((Cout + tn - 1) / tn) * ((Cin + tc - 1) / tc);
    if (const int v = ka_opt(KA_OPT_WGRAD_WGS, 0); v > 0) target_wgs = v;   // experiments
    int s = (target_wgs > 0 ? target_wgs : 256) / tiles;
    if (s < 1) s = 1;
    if (s > B) s = B;
    const int bps = (B + s - 1) / s;
    return (B + bps - 1) / bps;
}

// number of partial slabs the caller must provide room for ([splits][9][Cout][Cin] floats)
extern "C" int ka_wgrad_splits(int B, int Cin, int Cout, int target_wgs) { return wgrad_splits_for(B, Cin, Cout, target_wgs); }

template <bool FUSED>
static int launch_wgrad_flat(const WgradArgs& a, dim3 grid, hipStream_t st) {
    const size_t lds = 3 * (82 * (128 * 2 + 32) + 11 * 17 * (kTC * 2 + 32)) + 82 * 4;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&wgrad_flat_kernel<FUSED>), attr_done, "wgrad (lean)")) return rc;
    hipLaunchKernelGGL((wgrad_flat_kernel<FUSED>), grid, dim3(512), lds, st, a);
    return ka_check_launch("wgrad (lean)");
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
    const int nsplit = wgrad_splits_for(B, Cin, Cout, target_wgs);
    const int bps = (B + nsplit - 1) / nsplit;
    const int ntn = (Cout + tn - 1) / tn, ntiles = ntn * ((Cin + kTC - 1) / kTC);
    WgradArgs a{dy, x, in_scale, in_shift, in_bias, slab, B, Cin, Cout, relu, bps, ntn, ntiles, nsplit, ka_opt(KA_OPT_WGRAD_STAG, 1), ka_debug_stamps().load()};
    dim3 grid(8 * ntiles * ((nsplit + 7) / 8));
    int rc;
    const bool fused = in_scale || in_bias || relu;
#define KA_WG(T_, TN_) (fused ? launch_wgrad<T_, TN_, true>(a, grid, st) : launch_wgrad<T_, TN_, false>(a, grid, st))
    // (the lean kernel addresses a tensor through a 32-bit buffer descriptor with the board as a signed scalar offset)
    const bool fits32 = (unsigned long long)B * KA_BOARD * 2 * (unsigned long long)(Cin > Cout ? Cin : Cout) < 0x7fffffffull;
    if (dtype == KA_DTYPE_BF16 && tn == 128 && fits32 && ka_opt(KA_OPT_WGRAD_LEAN, 1) != 0)
        rc = fused ? launch_wgrad_flat<true>(a, grid, st) : launch_wgrad_flat<false>(a, grid, st);
    else if (dtype == KA_DTYPE_BF16) rc = tn == 64 ? KA_WG(bf16_t, 64) : KA_WG(bf16_t, 128);
    else if (dtype == KA_DTYPE_F32) rc = tn == 64 ? KA_WG(float, 64) : KA_WG(float, 128);
#undef KA_WG
    else { ka_set_error("wgrad: unknown dtype %d", dtype); return KA_ERR_ARG; }
    if (rc) return rc;
    const size_t total = (size_t)9 * Cout * (Cin / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, dw, nsplit, Cout, Cin, Cin_real,
                       accumulate);
    return ka_check_launch("wgrad_reduce");
}
