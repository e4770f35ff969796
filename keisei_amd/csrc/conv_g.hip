// 3x3 board convolution as a GEMM-class kernel ("conv_g"): 256-row x 256-channel output tiles, weights through LDS.
//
// Why a second conv kernel.  conv3x3_kernel (one board = 96 padded rows per 256-thread workgroup, each wave streaming its
// own 64-channel weight fragments from L2 into registers) is paced by three things its structure cannot remove: 15.6 % of
// its MFMAs multiply the 81 -> 96 padding, every workgroup pulls the whole 1.18 MB weight tensor through the L2 -> CU
// path (4.8 TB per launch, 13 TB/s, close to what that path sustains), and a vector-memory prefetch of the next staging
// step stalls the weight stream because vector-memory results return in order.  Here (bf16, Cin = Cout = 256):
//
//   * one persistent 512-thread workgroup per CU walks a contiguous range of boards in tiles of 3 boards = 243 rows packed
//     flat into 16 MFMA row tiles (5 % padding), all 256 output channels; 8 waves = 4 (row quarters) x 2 (channel halves),
//     wave tile 64 rows x 128 channels = 128 accumulator VGPRs.  A 1-board remainder tile deals its 6 row tiles round-robin
//     to the four row groups and skips the upper half of every wave's MFMAs (half a tile time);
//   * the activations of the tile's boards are staged as zero-haloed 17-wide images in 64-channel chunks (the 9 taps are
//     constant LDS offsets, as in conv3x3_kernel), with the fused input transform of the launch kind applied on the way
//     in; the pieces of the NEXT chunk (or of the next tile's first chunk) are requested 15 k-steps ahead and committed at
//     the chunk boundary -- a staging step costs two barriers and four LDS stores per thread;
//   * the weights of one (tap, 32-channel) k-step are ONE contiguous 16 KiB block of the fragment-ordered pack; they
//     travel L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 2 instructions per wave and k-step) through a ring of four
//     slots, three k-steps ahead, continuously across chunks and tiles, and all eight waves read their fragments from the
//     slot: 8x less L2 -> CU weight traffic per MFMA, and an LDS latency instead of an L2 latency.  The deep ring is also
//     what lets the staging loads sit in the same in-order vector-memory queue without stalling the matrix cores;
//   * one raw s_barrier per k-step (32 MFMAs per wave), hand-counted vmcnt waits for the weight ring; operand fragments
//     are read half a step (weights) / one step (activations) ahead into alternating register sets;
//   * epilogue: 16-byte stores straight from the accumulators; per-board channel sums / sums of squares (BatchNorm
//     statistics, SE squeeze): DPP reduction over the 16 lanes of a row, per-wave partials through the free slot of the
//     weight ring, summed in a fixed order (deterministic).
//
// KIND (template parameter; the hand-counted vmcnt immediates and the register budget depend on it):
//   0  plain input                                   statistics epilogue      conv1 forward
//   1  relu(x*scale + shift) + bias[b]               statistics epilogue      conv2 forward
//   2  x*scale + shift + in2*k3 (-> in_out)          masked epilogue          conv2 data gradient
//   3  x*scale + shift + in2*k3 (-> in_out)          plain store              conv1 data gradient
#include <stdlib.h>
#include "conv_g.h"

namespace {

constexpr int kPW = 17;                              // squares per padded board row (as conv3x3.hip)
constexpr int kImgSq = 10 * kPW + 11;                // 181 squares: what any tap of any board square can address
constexpr int kC = 256;
constexpr int kKC = 64;                              // channels per LDS chunk
constexpr int kStride = kKC * 2 + 32;                // bytes per square (+32: conflict-free fragment reads)
constexpr int kImgBytes = kImgSq * kStride;          // 28,960
constexpr int kTB = 3;                               // boards per full tile
constexpr int kSlot = 16 * 1024;                     // one k-step of weights: 16 channel tiles x 1 KiB
constexpr int kRing = 4;
constexpr int kOffRing = kTB * kImgBytes;            // 86,880
constexpr int kOffCoef = kOffRing + kRing * kSlot;   // [5][256] floats: scale | shift | k3 | ep_scale | ep_shift
constexpr int kOffBias = kOffCoef + 5 * kC * 4;      // [4][256] floats: per-board bias of the tile (KIND 1; row 3 is padding)
constexpr int kLds = kOffBias + 4 * kC * 4;          // 161,632 B
static_assert(kLds <= 160 * 1024, "conv_g LDS budget");

__device__ __forceinline__ int img_square(int p) { return (p / 9 + 1) * kPW + (p % 9) + 1; }
// channel slot permutation of the pack (conv3x3.hip chan_of with NT = 16: every tile has a partner)
__device__ __forceinline__ int chan_of16(int nt, int s) { return (nt >> 1) * 32 + (s >> 2) * 8 + (nt & 1) * 4 + (s & 3); }
__device__ __forceinline__ float row_sum16g(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define KA_BARRIER()                                          \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                         \
        asm volatile("" ::: "memory");                        \
    } while (0)

// Registers v240..v255 are RESERVED (amdgpu_num_vgpr(240) keeps the compiler in v0..v239): the staging pieces are loaded
// into them by inline asm and stay there, untouched by the register allocator, until the commit 15 k-steps later.  (As
// ordinary loads the compiler waits vmcnt(0) at the commit, draining the weight ring; as inline-asm loads into
// compiler-allocated registers it spills / reuses the destination before the data has arrived.)
template <int KIND>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(240))) void conv_g_kernel(ConvGArgs a) {
    constexpr bool IN2 = KIND >= 2, MASKED = KIND == 2, STATS = KIND <= 1;
    // vector-memory operations per staging request: 4 pieces (+ 4 of the second tensor | + 2 LDS-DMA words of the bias table)
    constexpr int P = IN2 ? 8 : (KIND == 1 ? 6 : 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + kOffRing;
    float* coef = reinterpret_cast<float*>(smem + kOffCoef);
    float* btab = reinterpret_cast<float*>(smem + kOffBias);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;          // row quarter (0..3), channel half (0, 1)

    // this workgroup's boards
    const int per = (a.B + gridDim.x - 1) / gridDim.x;
    const int bbeg = blockIdx.x * per, bend = min(a.B, bbeg + per);
    if (bbeg >= bend) return;

    // zero the three haloed images once (staging writes interior squares only); coefficient tables
    for (int i = tid; i < kTB * kImgBytes / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
    if (tid < kC) {
        coef[tid] = a.in_scale ? a.in_scale[tid] : 1.f;
        coef[kC + tid] = a.in_shift ? a.in_shift[tid] : 0.f;
        coef[2 * kC + tid] = IN2 ? a.in_k3[tid] : 0.f;
        coef[3 * kC + tid] = MASKED ? a.ep_scale[tid] : 0.f;
        coef[4 * kC + tid] = MASKED ? a.ep_shift[tid] : 0.f;
    }

    const char* wlane = static_cast<const char*>(a.wpack) + lane * 16;
    // weight block of k-step g (mod 72) of a tile: chunk kc = g / 18, tap = (g % 18) / 2, ks = kc * 2 + (g & 1)
    auto dma = [&](int G) {          // this wave's two 1 KiB pieces of global k-step G into slot G % 4
        const int g = G % 72, kc = g / 18, s = g - kc * 18, tap = s >> 1, ks = kc * 2 + (s & 1);
        const char* src = wlane + ((size_t)(tap * 8 + ks) * 16 + wave * 2) * 1024;
        char* dst = ring + (G & (kRing - 1)) * kSlot + (wave * 2) * 1024;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 1024, (lds_ptr)(dst + 1024), 16, 0, 0);
    };

    // ---- staging.  Rows of a tile: flat R = board * 81 + square; piece id = tid + 512 u -> row id / 8, 16-byte piece
    // id % 8 (= tid % 8: a thread always owns the same 8 channels of a chunk).
    const int spc = tid & 7;
    auto src_off = [&](int u, int b0, int nb, int kc) {
        const int row = min((tid + 512 * u) >> 3, nb * KA_BOARD - 1);   // clamped: rows past the tile are never committed
        return ((size_t)(b0 * KA_BOARD + row) * kC + kc * kKC + spc * 8) * 2;
    };
#define KA_LDR(reg_, p_) asm volatile("global_load_dwordx4 " reg_ ", %0, off" :: "v"(p_) : "memory", "v240", "v241", "v242", "v243", \
                                      "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255")
#define KA_REQUEST(b0_, nb_, kc_)                                                                                  \
    do {                                                                                                          \
        const char* in_ = static_cast<const char*>(a.in);                                                         \
        KA_LDR("v[240:243]", in_ + src_off(0, b0_, nb_, kc_)); KA_LDR("v[244:247]", in_ + src_off(1, b0_, nb_, kc_)); \
        KA_LDR("v[248:251]", in_ + src_off(2, b0_, nb_, kc_)); KA_LDR("v[252:255]", in_ + src_off(3, b0_, nb_, kc_)); \
        if (KIND == 1) {   /* bias rows of the tile's boards straight into the LDS table by LDS-DMA: 2 words per lane */ \
            _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                                     \
                const int e_ = (wave * 2 + k_) * 64 + lane;        /* table entry: board e_ / 256, channel e_ % 256 */ \
                const float* g_ = a.in_bias + (size_t)min((b0_) + (e_ >> 8), a.B - 1) * kC + (e_ & 255);          \
                __builtin_amdgcn_global_load_lds(g_, (lds_ptr)(btab + (wave * 2 + k_) * 64), 4, 0, 0);            \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
    // commit of piece u (reserved registers v[240 + 4u : 243 + 4u]); only behind a wait that covers the request
    auto lds_addr = [&](int row) {
        const int bd = row / KA_BOARD, sq = row - bd * KA_BOARD;
        return (unsigned)(size_t)(lds_ptr)(smem + bd * kImgBytes + img_square(sq) * kStride + spc * 16);
    };
#define KA_COMMIT_ONE(u_, r0_, r1_, r2_, r3_, rng_, b0_, nb_, kc_)                                                 \
    do {                                                                                                          \
        const int row_ = (tid + 512 * (u_)) >> 3;                                                                 \
        if (row_ < (nb_) * KA_BOARD) {                                                                            \
            if (KIND == 0) {                                                                                      \
                asm volatile("ds_write_b128 %0, " rng_ :: "v"(lds_addr(row_)) : "memory");                        \
            } else {                                                                                              \
                u32x4 v_;                                                                                         \
                asm volatile("v_mov_b32 %0, " r0_ "\n\tv_mov_b32 %1, " r1_ "\n\tv_mov_b32 %2, " r2_ "\n\tv_mov_b32 %3, " r3_ \
                             : "=v"(v_.x), "=v"(v_.y), "=v"(v_.z), "=v"(v_.w));                                   \
                bf16x8 x_ = __builtin_bit_cast(bf16x8, v_);                                                       \
                const int bd_ = row_ / KA_BOARD, c0_ = (kc_) * kKC + spc * 8;                                     \
                _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_)                                                   \
                    x_[e_] = (__bf16)(fmaxf(fmaf((float)x_[e_], coef[c0_ + e_], coef[kC + c0_ + e_]), 0.f) + btab[bd_ * kC + c0_ + e_]); \
                *reinterpret_cast<bf16x8*>(smem + bd_ * kImgBytes + img_square(row_ - bd_ * KA_BOARD) * kStride + spc * 16) = x_; \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define KA_COMMIT(b0_, nb_, kc_)                                                                                   \
    do {                                                                                                          \
        KA_COMMIT_ONE(0, "v240", "v241", "v242", "v243", "v[240:243]", b0_, nb_, kc_);                            \
        KA_COMMIT_ONE(1, "v244", "v245", "v246", "v247", "v[244:247]", b0_, nb_, kc_);                            \
        KA_COMMIT_ONE(2, "v248", "v249", "v250", "v251", "v[248:251]", b0_, nb_, kc_);                            \
        KA_COMMIT_ONE(3, "v252", "v253", "v254", "v255", "v[252:255]", b0_, nb_, kc_);                            \
    } while (0)

    // ---- row tiles of this wave: full tile (2 or 3 boards): row tiles wm*4 .. wm*4+3; 1-board tile: i*4 + wm, i < 2
    int rowoff[4];
    auto tile_of = [&](int i, int nb) { return nb == 1 ? i * 4 + wm : wm * 4 + i; };
    auto set_rows = [&](int nb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int R = tile_of(i, nb) * 16 + r;
            if (R >= nb * KA_BOARD) R = 0;                            // padding rows read valid LDS, are never stored
            rowoff[i] = (R / KA_BOARD) * kImgBytes + img_square(R % KA_BOARD) * kStride + q * 16;
        }
    };
    auto read_a = [&](bf16x8 (&af)[4], int s) {
        const int tap = s >> 1;
        const int aoff = ((tap / 3 - 1) * kPW + (tap % 3 - 1)) * kStride + (s & 1) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + rowoff[i] + aoff);
    };
    auto read_b = [&](bf16x8 (&bw)[4], int G, int half) {
        const char* slot = ring + (G & (kRing - 1)) * kSlot + (wn * 8 + half * 4) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[j] = *reinterpret_cast<const bf16x8*>(slot + j * 1024);
    };

    // ---- prologue: first tile's chunk 0 staged synchronously, weight ring primed three k-steps deep
    int b0 = bbeg, nb = min(kTB, bend - bbeg);
    int G = 0;
    KA_REQUEST(b0, nb, 0);
    dma(0); dma(1); dma(2);
    __syncthreads();                                                   // halo zeroing + coefficient tables done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (bias table by DMA, first ring slots)
    KA_BARRIER();
    KA_COMMIT(b0, nb, 0);
    KA_BARRIER();

    for (;;) {
        const bool half = nb == 1;                                    // 1-board tile: every wave owns at most 2 row tiles
        const int nxt_b0 = b0 + nb, nxt_nb = min(kTB, bend - nxt_b0); // next tile (nxt_nb <= 0: none)
        set_rows(nb);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 af0[4], af1[4], bw0[4], bw1[4];
        read_a(af0, 0);
        read_b(bw0, G, 0);
        // one k-step; AC = activation fragments of this step, AN = the set that receives the next step's
        auto step = [&](int g, bf16x8 (&AC)[4], bf16x8 (&AN)[4]) {
            const int kc = g / 18, s = g - kc * 18;
            const bool boundary = s == 17 && kc < 3;                  // the next chunk replaces the image inside this step
            // DMA(G+1) must have landed (its fragments are read in the second half of this step).  Younger than it:
            // DMA(G+2) and, when the next staging request was issued one or two steps ago, its P operations.
            if (s == 3 || s == 4) {
                if (P == 8) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (P == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            dma(G + 3);
            if (s == 2) {                                             // the next staging step's pieces: in flight for 16 k-steps
                if (kc < 3) KA_REQUEST(b0, nb, kc + 1);
                else if (nxt_nb > 0) KA_REQUEST(nxt_b0, nxt_nb, 0);
                else KA_REQUEST(b0, nb, 3);                           // (keeps the operation count fixed; never committed)
            }
            read_b(bw1, G, 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw0[j], AC[i], acc[i][j], 0, 0, 0);
            if (!half) {
#pragma unroll
                for (int i = 2; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw0[j], AC[i], acc[i][j], 0, 0, 0);
            }
            if (boundary) {                                           // requested at s == 2: long complete
                KA_BARRIER();                                          // every wave holds its copy of this step's fragments
                KA_COMMIT(b0, nb, kc + 1);
                KA_BARRIER();
            }
            // next step's first-half weights and activations (across a tile boundary the activation fragments read here
            // are discarded)
            read_b(bw0, G + 1, 0);
            read_a(AN, (g + 1) % 18);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw1[j], AC[i], acc[i][4 + j], 0, 0, 0);
            if (!half) {
#pragma unroll
                for (int i = 2; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw1[j], AC[i], acc[i][4 + j], 0, 0, 0);
            }
            ++G;
        };
        for (int g = 0; g < 72; g += 2) {
            step(g, af0, af1);
            step(g + 1, af1, af0);
        }

        // ---- epilogue.  Lane (r, q) of accumulator (i, j): row tile t(i), row R = 16 t + r, channels
        // chan_of16(wn*8 + j, 4q) .. +3; the tile pair (j, j+1), j even, gives 8 consecutive channels.
        // Per-wave statistics partials go through the free slot of the weight ring (slot (G+3) % 4 = the one read last):
        // part[statistic][wm][slot 0/1][256 channels]; a wave's rows lie in at most two boards, kbase and kbase + 1.
        float* part = reinterpret_cast<float*>(ring + ((G + 3) & (kRing - 1)) * kSlot);
        const int kbase = nb == kTB ? (wm * 64) / KA_BOARD : 0;
        if (STATS) KA_BARRIER();                                       // every wave has finished reading that slot
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int cb = chan_of16(wn * 8 + 2 * jp, 4 * q);          // first of this lane's 8 channels
            float s0[2][8], s1[2][8];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[k][e] = 0.f; s1[k][e] = 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = tile_of(i, nb) * 16 + r;
                const bool valid = R < nb * KA_BOARD && (i < 2 || !half);
                const bool hi = R / KA_BOARD != kbase;                 // per lane: a row tile may straddle two boards
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e - 4];
                    o[e] = (__bf16)v;
                    if (STATS) {
                        const float vv = valid ? v : 0.f;
                        s0[0][e] += hi ? 0.f : vv; s0[1][e] += hi ? vv : 0.f;
                        s1[0][e] += hi ? 0.f : vv * vv; s1[1][e] += hi ? vv * vv : 0.f;
                    }
                }
                if (valid)
                    *reinterpret_cast<bf16x8*>(static_cast<char*>(a.out) + ((size_t)(b0 * KA_BOARD + R) * kC + cb) * 2) = o;
            }
            if (STATS) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s0[k][e] = row_sum16g(s0[k][e]); s1[k][e] = row_sum16g(s1[k][e]); }
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        float* p0 = part + ((0 * 4 + wm) * 2 + k) * kC + cb;
                        float* p1 = part + ((1 * 4 + wm) * 2 + k) * kC + cb;
                        *reinterpret_cast<f32x4*>(p0) = f32x4{s0[k][0], s0[k][1], s0[k][2], s0[k][3]};
                        *reinterpret_cast<f32x4*>(p0 + 4) = f32x4{s0[k][4], s0[k][5], s0[k][6], s0[k][7]};
                        *reinterpret_cast<f32x4*>(p1) = f32x4{s1[k][0], s1[k][1], s1[k][2], s1[k][3]};
                        *reinterpret_cast<f32x4*>(p1 + 4) = f32x4{s1[k][4], s1[k][5], s1[k][6], s1[k][7]};
                    }
                }
            }
        }
        if (STATS) {
            KA_BARRIER();
            // board k of the tile, channel c: the partials of the waves whose rows touch board k, in wave order
            for (int o = tid; o < nb * kC; o += 512) {
                const int k = o >> 8, c = o & 255;
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int kb = nb == kTB ? (w * 64) / KA_BOARD : 0;
                    const int slot = k - kb;
                    if (slot == 0 || slot == 1) {
                        t0 += part[((0 * 4 + w) * 2 + slot) * kC + c];
                        t1 += part[((1 * 4 + w) * 2 + slot) * kC + c];
                    }
                }
                if (a.bsum) a.bsum[(size_t)(b0 + k) * kC + c] = t0;
                if (a.sqpart) a.sqpart[(size_t)(b0 + k) * kC + c] = t1;
            }
        }

        if (nxt_nb <= 0) break;
        // ---- next tile: its first chunk (requested at s == 2 of the last chunk) replaces the image
        b0 = nxt_b0; nb = nxt_nb;
        KA_BARRIER();                                                  // every wave is done with the old image / the partials
        KA_COMMIT(b0, nb, 0);
        KA_BARRIER();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // look-ahead weight pieces of three k-steps that never run
}

template <int KIND>
int conv_g_launch(const ConvGArgs& a, int grid, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv_g_kernel<KIND>), done, "conv_g")) return rc;
    hipLaunchKernelGGL(conv_g_kernel<KIND>, dim3(grid), dim3(512), kLds, st, a);
    return ka_check_launch("conv_g");
}

}  // namespace

// bf16, Cin = Cout = 256, a batch that fills the chip, and a launch kind this build covers (forward convolutions; the
// data-gradient kinds stay on conv3x3_kernel)
bool conv_g_applies(int B, int Cin, int Cout, int dtype, bool two_tensor_input) {
    const char* e = getenv("KA_CONV_G");                              // opt-in: conv3x3_kernel is the faster one
    if (!e || atoi(e) == 0) return false;
    return dtype == KA_DTYPE_BF16 && Cin == kC && Cout == kC && B >= 768 && !two_tensor_input;
}

int conv_g_run(const ConvGArgs& a, hipStream_t st) {
    KA_REQUIRE(a.in && a.wpack && a.out && a.B > 0, "conv3x3 (conv_g): null tensor");
    KA_REQUIRE(!a.in2, "conv_g: launch kind not built");
    int grid = 256;
    if (const char* e = getenv("KA_CONV_G_WGS")) { const int v = atoi(e); if (v > 0) grid = v; }
    if (grid > (a.B + kTB - 1) / kTB) grid = (a.B + kTB - 1) / kTB;
    if (a.in_scale || a.in_bias || a.relu) {
        KA_REQUIRE(a.in_scale && a.in_shift && a.in_bias && a.relu, "conv_g: the transformed-input form is relu(x*scale+shift)+bias");
        return conv_g_launch<1>(a, grid, st);
    }
    return conv_g_launch<0>(a, grid, st);
}
