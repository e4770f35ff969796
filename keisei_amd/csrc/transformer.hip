// Transformer encoder path (BASELINE config 5; reference keisei/training/models/transformer.py:37-95 =
// nn.TransformerEncoder(norm_first, batch_first, relu FFN, dropout 0.1) over the 81 board squares).
//
//   tokens (B*81, d) row-major, activations bf16 (AMP) or fp32 (parity mode)
//
// Kernels here:
//   * gemm_nt_bf16_kernel   C = A * B^T on v_mfma_f32_16x16x32_bf16 (both operands K-contiguous), 128x128x64 LDS tiles,
//                           register prefetch of the next K tile; epilogue bias / ReLU / dropout / residual; split-K
//                           slabs.  Forward (B = weight), input gradient (B = transposed weight copy) and weight
//                           gradient (A, B = transposed activations) are all this one form -- the transposed bf16
//                           copies are produced by transpose_pad_kernel (activations) / the weight cache refresh.
//   * attention_{fwd,bwd}   one wave per (board, head): Q, K, V (81 x dh) live in LDS, scores on the matrix cores
//                           (bf16 MFMA, or the exact-f32 MFMA in parity mode), softmax in registers (row = 4 registers x
//                           16 lanes), P only ever exists as 16/32-row tiles; backward recomputes P from the saved
//                           log-sum-exp.
//   * layer norm forward / backward, positional embedding add / gradient, dropout (counter-based hash: the mask is
//     recomputed, never stored), mean pool, tanh.
// The fp32 parity mode runs its linear layers on ka_gemm (gemm.hip, exact-f32 MFMA).
#include "common.h"

namespace {

// ------------------------------------------------------------------ dropout: counter-based keep mask
// keep(element) = 16 bits of hash32(seed, index >> 1) >= p * 2^16: one 32-bit hash (two multiplies) serves the two
// elements of an index pair, so the epilogues and elementwise kernels that own runs of consecutive elements pay half a
// hash per element (the 64-bit mixer used before cost more than the 16 MFMAs of a 128x128x32 GEMM step per output
// tile).  Recomputed wherever it is needed (forward epilogues, backward); never stored.  p is realised to 2^-16.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_thresh(float p) { return p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u; }
// the same mask for indices below 2^32, on 32-bit arithmetic (key = drop_key(seed), computed once)
__device__ __forceinline__ uint32_t drop_key(unsigned long long seed) { return (uint32_t)seed * 0x9E3779B1U + (uint32_t)(seed >> 32) * 0x85EBCA6BU; }
__device__ __forceinline__ float keep_scale32(uint32_t key, uint32_t index, uint32_t thresh, float inv_keep) {
    const uint32_t h = hash32((index >> 1) * 0x9E3779B1U + key);
    return ((index & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thresh ? inv_keep : 0.f;
}
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long index, uint32_t thresh, float inv_keep) {
    const unsigned long long pair = index >> 1;
    const uint32_t key = (uint32_t)seed * 0x9E3779B1U + (uint32_t)(seed >> 32) * 0x85EBCA6BU;      // uniform
    const uint32_t h = hash32((uint32_t)pair * 0x9E3779B1U + (uint32_t)(pair >> 32) * 0xC2B2AE35U + key);
    const uint32_t bits = (index & 1) ? (h >> 16) : (h & 0xFFFFu);
    return bits >= thresh ? inv_keep : 0.f;
}

template <typename T> __device__ __forceinline__ float ldT(const T* p, size_t i);
template <> __device__ __forceinline__ float ldT<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldT<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i].v); }
template <typename T> __device__ __forceinline__ void stT(T* p, size_t i, float v);
template <> __device__ __forceinline__ void stT<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stT<bf16_t>(bf16_t* p, size_t i, float v) { p[i].v = f2bf(v); }

// ------------------------------------------------------------------ NT GEMM on the bf16 matrix cores
struct NtArgs {
    const uint16_t* A; const uint16_t* B;     // bf16 [M][lda], [N][ldb], K-contiguous, K % 32 == 0 (zero padded)
    void* C; const float* bias; const void* residual;     // C [M][ldc] bf16 or fp32; residual like C (same dtype as C)
    const void* relu_act;                      // bf16 [M][ldc] or null: the output is zeroed where relu_act <= 0 (ReLU backward; bf16 C only)
    int M, N, K, lda, ldb, ldc;
    int c_bf16, relu, ksplit_len;              // ksplit_len < K: gridDim.z slabs of fp32 [z][M][ldc], no epilogue
    int lds_epilogue;                          // full bf16 tiles leave through LDS as whole-row 16-byte pieces (KA_TF_LDS_EPI=0: off)
    float drop_p; unsigned long long seed;     // dropout on the (bias, relu)'d value before the residual add
    int row0;                                  // gemm_nt_k256_kernel: first row of this launch's panels (the ragged last panel is its own launch)
    int map_gx, map_gy;                        // gemm_nt_bf16_kernel: > 0 = one-dimensional grid walked as 8 x 8 super-tiles per XCD (n-tiles, m-tiles)
};

constexpr int kBM = 128, kBN = 128, kBK = 64, kLdsStride = kBK * 2 + 16;     // bytes per tile row (144: 16 rows x 16 B land on 64 different banks)

// 128x128x64 tiles, 256 threads (wave = 64x64 of the tile: 16 accumulator tiles), register-staged double buffering: the
// global loads of k-tile t+1 are in flight during the 32 MFMAs of k-tile t, one barrier per k-tile.  K is a multiple of
// 32; a trailing half tile is zero-filled.  Epilogue: a lane owns 4 consecutive columns of a row in every accumulator
// tile, so bias / residual / output move as 16- or 8-byte pieces when N and ldc allow it (vec4).
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(NtArgs g) {
    extern __shared__ __attribute__((aligned(16))) char nt_smem[];
    auto As = [&](int buf) { return nt_smem + buf * (kBM * kLdsStride); };
    auto Bs = [&](int buf) { return nt_smem + 2 * kBM * kLdsStride + buf * (kBN * kLdsStride); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile map: workgroups are dealt round-robin over the 8 XCDs (linear id L sits on XCD L % 8), each with its own
    // L2.  The n-tiles of one m-tile read the same 128 rows of A: they are given consecutive slots of ONE XCD, so A comes
    // from HBM once instead of once per XCD that happens to run one of its n-tiles (N = 1024: 8 n-tiles).  Placement only.
    // Many n-tiles AND many m-tiles (the policy layer: 88 x 32 tiles over K = 20 736): neither operand panel fits a cache, and
    // with one m-tile per XCD at a time every m-tile streams the whole weight matrix through its L2 (measured: 15.6 GB of L2
    // misses per launch for 0.64 GB of operands, 5.4 TB/s -- the fabric, not the matrix pipe, paced it).  There the grid is
    // one-dimensional and an XCD walks 8 x 8 super-tiles: the 64 workgroups resident on it share 8 A and 8 B panels.
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.map_gx > 0) {
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int sm_n = (g.map_gy + 7) >> 3;                // super-tiles along m
        const int G = (slot >> 6) * 8 + xcd, within = slot & 63;
        by = (G % sm_n) * 8 + (within & 7);
        bx = (G / sm_n) * 8 + (within >> 3);
        if (by >= g.map_gy || bx >= g.map_gx) return;        // padding of the last super-tiles (uniform exit, no barrier yet)
    } else {
        const int gx = gridDim.x, gy = gridDim.y, L = by * gx + bx;
        const int full = (gy / 8) * 8;                       // m-tiles covered by whole groups of 8
        const int xcd = L & 7, slot = L >> 3;
        const int mt = (slot / gx) * 8 + xcd;
        if (L < full * gx && mt < full) { by = mt; bx = slot % gx; }
    }
    const int m0 = by * kBM, n0 = bx * kBN;
    const int kbeg = blockIdx.z * g.ksplit_len, kend = min(g.K, kbeg + g.ksplit_len);
    // staging role: 128 rows x 8 pieces (16 B) per operand tile = 1024 pieces, four per thread and operand
    const int srow = tid >> 3, spc = tid & 7;
    uint4 ra[4], rb[4];
    auto load = [&](int k0) {
        const bool kok = k0 + spc * 8 < kend;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = srow + 32 * h;
            const int m = m0 + row, n = n0 + row;
            ra[h] = (kok && m < g.M) ? *reinterpret_cast<const uint4*>(g.A + (size_t)m * g.lda + k0 + spc * 8) : uint4{0, 0, 0, 0};
            rb[h] = (kok && n < g.N) ? *reinterpret_cast<const uint4*>(g.B + (size_t)n * g.ldb + k0 + spc * 8) : uint4{0, 0, 0, 0};
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = srow + 32 * h;
            *reinterpret_cast<uint4*>(As(buf) + row * kLdsStride + spc * 16) = ra[h];
            *reinterpret_cast<uint4*>(Bs(buf) + row * kLdsStride + spc * 16) = rb[h];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) {
        load(kbeg);
        store(0);
        __syncthreads();
        int buf = 0;
        for (int k0 = kbeg; k0 < kend; k0 += kBK) {
            const bool more = k0 + kBK < kend;
            if (more) load(k0 + kBK);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *reinterpret_cast<const bf16x8*>(As(buf) + (wm * 64 + i * 16 + r) * kLdsStride + kk * 64 + q * 16);
                    bfr[i] = *reinterpret_cast<const bf16x8*>(Bs(buf) + (wn * 64 + i * 16 + r) * kLdsStride + kk * 64 + q * 16);
                }
                // C^T tiles: the weight-like operand B is the MFMA "A" so that a lane holds 4 consecutive columns n of one row m
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            if (more) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue: lane (r, q) of tile (i, j): row m = i*16 + r, columns n = j*16 + 4q .. +3
    const bool split = g.ksplit_len < g.K;
    const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(g.drop_p);
    const bool vec4 = (g.N & 3) == 0 && (g.ldc & 3) == 0;      // (the C / residual / bias bases are whole allocations)
    // full bf16 tiles with 8-element alignment: the accumulators go through LDS (the operand tiles are dead) so that a thread
    // owns 8 consecutive columns of a row and 16 lanes write a whole 256-byte row piece -- one 16-byte store per piece
    // instead of 32-byte pieces scattered over 16 rows per instruction; bias / residual likewise arrive as 16 / 32-byte pieces
    if (!split && g.c_bf16 && (g.N & 7) == 0 && (g.ldc & 7) == 0 && m0 + kBM <= g.M && n0 + kBN <= g.N && g.lds_epilogue) {
        constexpr int kEs = kBN * 4 + 16;                       // 528-byte rows: the 8-lane groups of a b128 write hit distinct banks
        __syncthreads();                                        // every wave is done with the operand tiles
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(nt_smem + (wm * 64 + i * 16 + r) * kEs + (wn * 64 + j * 16 + 4 * q) * 4) = acc[i][j];
        __syncthreads();
        const int pc8 = tid & 15, row0 = tid >> 4;              // 16 pieces of 8 columns per row, 16 rows per pass
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) { b0 = *reinterpret_cast<const f32x4*>(g.bias + n0 + pc8 * 8); b1 = *reinterpret_cast<const f32x4*>(g.bias + n0 + pc8 * 8 + 4); }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = row0 + 16 * pass;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(nt_smem + row * kEs + pc8 * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(nt_smem + row * kEs + pc8 * 32 + 16);
            float v[8] = {v0[0] + b0[0], v0[1] + b0[1], v0[2] + b0[2], v0[3] + b0[3], v1[0] + b1[0], v1[1] + b1[1], v1[2] + b1[2], v1[3] + b1[3]};
            const size_t ob = (size_t)(m0 + row) * g.ldc + n0 + pc8 * 8;
            if (g.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (g.drop_p > 0.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= keep_scale(g.seed, ob + e, thresh, inv_keep);
            }
            if (g.relu_act) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(g.relu_act) + ob);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)av[e] > 0.f ? v[e] : 0.f;
            }
            if (g.residual) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(g.residual) + ob);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(static_cast<uint16_t*>(g.C) + ob) = o;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + 4 * q;
            if (nb >= g.N) continue;
            const size_t ob = (size_t)m * g.ldc + nb;
            if (vec4) {                                          // nb + 3 < N
                f32x4 v = acc[i][j];
                if (split) { *reinterpret_cast<f32x4*>(static_cast<float*>(g.C) + (size_t)blockIdx.z * g.M * g.ldc + ob) = v; continue; }
                if (g.bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + nb); v += bv; }
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (g.drop_p > 0.f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= keep_scale(g.seed, ob + e, thresh, inv_keep);
                }
                if (g.c_bf16) {
                    if (g.relu_act) {
                        const bf16x4 av = *reinterpret_cast<const bf16x4*>(static_cast<const uint16_t*>(g.relu_act) + ob);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (float)av[e] > 0.f ? v[e] : 0.f;
                    }
                    if (g.residual) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(static_cast<const uint16_t*>(g.residual) + ob);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(static_cast<uint16_t*>(g.C) + ob) = o;
                } else {
                    if (g.residual) v += *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.residual) + ob);
                    *reinterpret_cast<f32x4*>(static_cast<float*>(g.C) + ob) = v;
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nb + e;
                if (n >= g.N) continue;
                float v = acc[i][j][e];
                const size_t o = ob + e;
                if (split) { static_cast<float*>(g.C)[(size_t)blockIdx.z * g.M * g.ldc + o] = v; continue; }
                if (g.bias) v += g.bias[n];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.drop_p > 0.f) v *= keep_scale(g.seed, o, thresh, inv_keep);
                if (g.c_bf16) {
                    if (g.relu_act && !(bf2f(static_cast<const uint16_t*>(g.relu_act)[o]) > 0.f)) v = 0.f;
                    if (g.residual) v += bf2f(static_cast<const uint16_t*>(g.residual)[o]);
                    static_cast<uint16_t*>(g.C)[o] = f2bf(v);
                } else {
                    if (g.residual) v += static_cast<const float*>(g.residual)[o];
                    static_cast<float*>(g.C)[o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ NT GEMM, 256 x 256 tiles: the policy layer
// M, N and K all large (policy_fc: 4096 x 11 259 x 20 736 forward, and its two gradients): with 128 x 128 tiles every
// workgroup pulls 256 operand rows per 64 k through the L2 for 32 MFMAs per wave -- 10.7 TB/s of L2 -> CU traffic at
// 0.7 PFLOP/s, matrix pipe 30-34 % busy.  Here a 512-thread workgroup owns a 256 x 256 tile: 8 waves as 2 (m) x 4 (n), wave
// tile 128 x 64 = 32 accumulator tiles (128 registers), 12 fragment reads per 32 MFMAs instead of 16, half the operand
// bytes per FLOP through L2 and LDS.  One workgroup per CU; an XCD walks 4 (m) x 8 (n) super-tiles (its 32 resident
// workgroups share 4 A and 8 B panels).  Operands reach the LDS by DMA, three 32-deep k-tiles ahead (below): matrix pipe
// 45-47 % busy (37 % with one register-staged 64-deep tile in flight and a drain at every barrier).  Bias-only epilogue, fp32
// or bf16 output, ragged M / N edges; K a multiple of 32.  Same products in the same k order as the 128 x 128 kernel:
// results are bit-identical.
constexpr int kGM = 256, kGN = 256;
constexpr int kGLds = 4 * (kGM + kGN) * 32 * 2;         // four stages of two unpadded [256][32] bf16 tiles: 128 KB

__global__ __launch_bounds__(512, 1) void gemm_nt_big_kernel(NtArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char big_smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // Four LDS stages of one 32-deep k-tile each ([256 + 256 rows][32 k] bf16 = 32 KB, unpadded 64-byte rows), filled by LDS-DMA
    // (global_load_lds_dwordx4: a wave instruction writes 64 lanes x 16 B = sixteen whole rows) three tiles ahead of the MFMAs,
    // the fragments of the NEXT tile read under the MFMAs of this one, a counted vmcnt before each barrier (never 0 in the loop)
    // and a raw s_barrier (a __syncthreads() would drain the DMA).
    // Bank conflicts are avoided on the SOURCE side: LDS slot s (16 B) of row R holds global piece s ^ (((R >> 2) & 1) << 1), which
    // puts the 16 lanes of every ds_read_b128 group on 16 different 16-byte bank slots.
    constexpr int BK = 32, kRow = BK * 2, kTile = kGM * kRow, kStage = 2 * kTile, NST = 4;      // 64 B, 16 KB, 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    int bx, by;
    {
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int sm_n = (g.map_gy + 3) >> 2;                // super-tiles along m
        const int G = (slot >> 5) * 8 + xcd, within = slot & 31;
        by = (G % sm_n) * 4 + (within & 3);
        bx = (G / sm_n) * 8 + (within >> 2);
        if (by >= g.map_gy || bx >= g.map_gx) return;        // padding of the last super-tiles (uniform exit, no barrier yet)
    }
    const int m0 = by * kGM, n0 = bx * kGN;
    // staging role: an operand tile is 16 pieces of 1 KB (16 rows); wave w issues pieces w and w + 8 of each operand.  Lane l of
    // piece p: row 16 p + (l >> 2), LDS slot l & 3 <- global piece (l & 3) ^ (((row >> 2) & 1) << 1).  Rows past the edge read the
    // operand's last row (their products land in accumulators that are never stored).
    uint32_t ao[2], bo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 16 * (wave + 8 * h) + (lane >> 2), pcs = (lane & 3) ^ (((row >> 2) & 1) << 1);
        ao[h] = (uint32_t)(min(m0 + row, g.M - 1) * g.lda + pcs * 8) * 2u;
        bo[h] = (uint32_t)(min(n0 + row, g.N - 1) * g.ldb + pcs * 8) * 2u;
    }
    auto dma = [&](int k0, int st) {
        const char* ak = reinterpret_cast<const char*>(g.A) + (size_t)k0 * 2;       // uniform
        const char* bk = reinterpret_cast<const char*>(g.B) + (size_t)k0 * 2;
        char* base = big_smem + st * kStage;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_global_load_lds(ak + ao[h], (lds_ptr)(base + (wave + 8 * h) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(bk + bo[h], (lds_ptr)(base + kTile + (wave + 8 * h) * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment address of lane (r, q): row r (+ 16 i), piece q -> slot q ^ (((r >> 2) & 1) << 1)   (16 | row offsets: same swizzle)
    const int foff = r * kRow + ((q ^ (((r >> 2) & 1) << 1)) << 4);
    const int nk = g.K / BK;
    struct Frags { bf16x8 a[8], b[4]; };
    auto rd = [&](int st, Frags& f) {
        const char* ab = big_smem + st * kStage + (wm * 128) * kRow + foff;
        const char* bb = big_smem + st * kStage + kTile + (wn * 64) * kRow + foff;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(bb + j * 16 * kRow);
#pragma unroll
        for (int i = 0; i < 8; ++i) f.a[i] = *reinterpret_cast<const bf16x8*>(ab + i * 16 * kRow);
    };
    // Invariant at the top of iteration t: tiles <= t+1 have landed and are visible to every wave, tile t's fragments are in
    // `cur`.  The iteration requests tile t+3, reads tile t+1's fragments under tile t's 32 MFMAs, then waits until at most the
    // newest tile's DMA (4 per lane) is outstanding -- tile t+2 has landed -- and passes the barrier that publishes it.  A stage is
    // written again (tile t+4 in iteration t+1) two barriers after its last read (iteration t-1).
    auto step = [&](int t, const Frags& cur, Frags& nxt) {
        if (t + 3 < nk) dma((t + 3) * BK, (t + 3) & (NST - 1));
        if (t + 1 < nk) rd((t + 1) & (NST - 1), nxt);
        // C^T tiles, as gemm_nt_bf16_kernel: a lane holds 4 consecutive columns n of one row m
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
        // issue order: the 12 fragment reads of the next tile spread over this tile's MFMAs (one read, then three MFMAs; the rest)
#pragma unroll
        for (int z = 0; z < 10; ++z) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                       // (nothing of the next iteration is moved in front of the barrier)
    };
    // prologue: tiles 0, 1, 2 on their way; 0 and 1 landed and published, tile 0's fragments read
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nk) dma(t * BK, t);
    if (nk >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frags fa, fb;
    rd(0, fa);
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        step(t, fa, fb);
        step(t + 1, fb, fa);
    }
    if (t < nk) step(t, fa, fb);
    // epilogue: lane (r, q) of tile (i, j): row m = i*16 + r, columns n = j*16 + 4q .. +3
    const bool vec4 = (g.N & 3) == 0 && (g.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + 4 * q;
            if (nb >= g.N) continue;
            const size_t ob = (size_t)m * g.ldc + nb;
            if (vec4) {
                f32x4 v = acc[i][j];
                if (g.bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + nb); v += bv; }
                if (g.c_bf16) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(static_cast<uint16_t*>(g.C) + ob) = o;
                } else {
                    *reinterpret_cast<f32x4*>(static_cast<float*>(g.C) + ob) = v;
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nb + e;
                if (n >= g.N) continue;
                float v = acc[i][j][e];
                if (g.bias) v += g.bias[n];
                if (g.c_bf16) static_cast<uint16_t*>(g.C)[ob + e] = f2bf(v);
                else static_cast<float*>(g.C)[ob + e] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ NT GEMM, K = 256: activations stationary in registers
// With d_model = 256 an output tile of gemm_nt_bf16_kernel has four k-tiles: prologue, four barriers and the epilogue of a
// 128 x 128 tile take 13.7 us for 1 us of MFMAs, and the N = 1024 products (FFN linear1, the input gradient of linear2) ran at
// 1.5 TB/s of their 850 MB (profiles/r03_transformer_by_shape.txt).  Here the contraction is turned inside out: a 256-thread
// workgroup owns 128 token rows and keeps them -- 32 rows x 256 k per wave = 16 operand fragments, 64 registers -- for its
// whole life, and walks ALL the output columns in chunks of 64: the chunk's weights (64 x 256, 32 KB from L2) sit in one of
// two LDS buffers, the next chunk is in flight in registers meanwhile, one barrier per chunk.  Per chunk and wave 32 LDS
// fragment reads feed 64 MFMAs; what leaves the kernel is the output, in 32-byte runs per lane (the weight rows of a chunk are
// permuted so that a lane owns 16 consecutive columns of its row).  Same products in the same k order as
// gemm_nt_bf16_kernel and the same epilogue statements: results are bit-identical (tests/test_hip_transformer.py).
constexpr int kKsK = 256, kKsCols = 64, kKsRows = 128, kKsStride = kKsK * 2 + 32;      // 544-byte weight rows: conflict-free ds_read_b128
constexpr int kKsLds = 2 * kKsCols * kKsStride;                                         // + N floats of bias behind the two buffers

// FULL: every row of the panel exists; RES / ACT: the epilogue reads a residual / a saved activation.  They are template
// parameters (and N is a multiple of 64, the bias sits in LDS) so that the NUMBER of vector-memory instructions per chunk is
// a compile-time constant: loads, stores and their waits share one in-order counter, and with a run-time count the compiler
// must wait `vmcnt(0)` for the next chunk's weights -- i.e. for the output stores just issued (measured: 4.1 us per chunk for
// 0.5 us of MFMAs).  With static counts the wait for the weights leaves the chunk's stores in flight.
template <bool FULL, bool RES, bool ACT>
__global__ __launch_bounds__(256, 2) void gemm_nt_k256_kernel(NtArgs g) {
    extern __shared__ __attribute__((aligned(16))) char ks_smem[];
    float* lbias = reinterpret_cast<float*>(ks_smem + kKsLds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int m0 = g.row0 + blockIdx.x * kKsRows + wave * 32;
    for (int i = tid; i < g.N; i += 256) lbias[i] = g.bias ? g.bias[i] : 0.f;
    // ---- this wave's 32 rows, all of K: X[rt][ks] = rows m0 + rt*16 + r, k = ks*32 + 8q .. +7
    bf16x8 X[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int m = m0 + rt * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            X[rt][ks] = (FULL || m < g.M) ? *reinterpret_cast<const bf16x8*>(g.A + (size_t)m * g.lda + ks * 32 + q * 8) : bf16x8{};
    }
    // ---- weight chunk staging: 64 rows x 32 pieces, eight per thread.  MFMA tile ct, row 4*qq + i of the chunk is output column
    // (ct >> 1)*32 + qq*8 + (ct & 1)*4 + i: lane qq then owns columns qq*8 .. +7 of each half of the chunk, and the four lanes
    // of a token row write one whole 64-byte sector per store instruction
    const int spc = tid & 31, srow0 = tid >> 5;                      // piece (16 B) and first of 8 rows (stride 8)
    // (macros, not lambdas over a shared array: hipcc kept such an array in scratch)
    const uint16_t* wsrc = g.B + (size_t)srow0 * g.ldb + spc * 8;
    uint32_t wdst[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const int nl = srow0 + 8 * h;                                // column within the chunk = hh*32 + qq*8 + c1*4 + i
        const int L = ((nl >> 5) * 2 + ((nl >> 2) & 1)) * 16 + ((nl >> 3) & 3) * 4 + (nl & 3);      // LDS row (hh*2 + c1)*16 + 4*qq + i
        wdst[h] = (uint32_t)(L * kKsStride + spc * 16);
    }
#define KA_WLOAD(W_, n0_)                                                                                       \
    _Pragma("unroll") for (int h = 0; h < 8; ++h)                                                               \
        W_[h] = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)((n0_) + 8 * h) * g.ldb);
#define KA_WSTORE(W_, buf_)                                                                                     \
    _Pragma("unroll") for (int h = 0; h < 8; ++h)                                                               \
        *reinterpret_cast<bf16x8*>(ks_smem + (buf_) * (kKsCols * kKsStride) + wdst[h]) = W_[h];
    const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(g.drop_p), key = drop_key(g.seed);
    const int nchunks = g.N / kKsCols;
    {
        bf16x8 w0[8];
        KA_WLOAD(w0, 0)
        KA_WSTORE(w0, 0)
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1, n0 = c * kKsCols;
        bf16x8 wr[8];                                                   // (a plain vector type: an array of HIP's uint4 structs stayed in scratch)
        KA_WLOAD(wr, min(c + 1, nchunks - 1) * kKsCols)              // (the last iteration re-reads its own chunk: a static count)
        // the epilogue's residual / activation pieces are requested before the MFMAs that hide their latency
        bf16x8 rres[RES ? 4 : 1], ract[ACT ? 4 : 1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = FULL ? m0 + rt * 16 + r : min(m0 + rt * 16 + r, g.M - 1);
                const size_t ob = (size_t)m * g.ldc + n0 + h * 32 + q * 8;
                if (RES) rres[RES ? rt * 2 + h : 0] = *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(g.residual) + ob);
                if (ACT) ract[ACT ? rt * 2 + h : 0] = *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(g.relu_act) + ob);
            }
        f32x4 acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* wb = ks_smem + buf * (kKsCols * kKsStride) + r * kKsStride + q * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + ct * 16 * kKsStride + ks * 64);
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, X[0][ks], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, X[1][ks], acc[1][ct], 0, 0, 0);
            }
        }
        // ---- epilogue: lane (r, q) holds, for row tile rt and half h of the chunk, columns n0 + h*32 + q*8 .. +7
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nb = n0 + h * 32 + q * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(lbias + nb), b1 = *reinterpret_cast<const f32x4*>(lbias + nb + 4);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int m = m0 + rt * 16 + r;
                const size_t ob = (size_t)m * g.ldc + nb;
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = acc[rt][2 * h][i] + b0[i]; v[4 + i] = acc[rt][2 * h + 1][i] + b1[i]; }
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (g.drop_p > 0.f) {
                    if ((unsigned long long)g.M * (unsigned long long)g.ldc <= 0xFFFFFFFFull) {      // (uniform) one 32-bit hash per element pair
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {             // ob is even: elements e, e + 1 share a hash
                            const uint32_t hsh = hash32((((uint32_t)ob + e) >> 1) * 0x9E3779B1U + key);
                            v[e] *= (hsh & 0xFFFFu) >= thresh ? inv_keep : 0.f;
                            v[e + 1] *= (hsh >> 16) >= thresh ? inv_keep : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= keep_scale(g.seed, ob + e, thresh, inv_keep);
                    }
                }
                if (ACT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)ract[ACT ? rt * 2 + h : 0][e] > 0.f ? v[e] : 0.f;
                }
                if (RES) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rres[RES ? rt * 2 + h : 0][e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                if (FULL || m < g.M) *reinterpret_cast<bf16x8*>(static_cast<uint16_t*>(g.C) + ob) = o;
            }
        }
        KA_WSTORE(wr, buf ^ 1)
        __syncthreads();
    }
#undef KA_WLOAD
#undef KA_WSTORE
}

// ------------------------------------------------------------------ TN GEMM: C[n][k] = sum_m A[m][n] * B[m][k]
// The weight gradient of a linear layer (dW = dY^T X, contraction over the tokens) straight from the row-major
// activations: [64 tokens][128 columns] tiles of both operands are staged row-major in LDS and the MFMA fragments are
// read with the hardware transpose read (ds_read_b64_tr_b16), as wgrad.hip does for the convolutions -- no transposed
// copy of either tensor (the NT form needed two transposes per layer: 8 % of a step).  128x128 output tile per
// 256-thread workgroup, split over token ranges (gridDim.z fp32 slabs, reduced in a fixed order by ka_reduce_slabs).
struct TnArgs {
    const uint16_t* A; const uint16_t* B; float* C;      // A [M][lda] (N columns used), B [M][ldb] (K columns used), C [z][N][ldc]
    int M, N, K, lda, ldb, ldc, msplit_len;
    float* colsum;                                       // [z][N] or null: column sums of A (the layer's bias gradient), see below
};
constexpr int kTnRows = 64, kTnStride = 128 * 2 + 32;    // 288 B: 8 consecutive rows of a transpose read spread over all banks
typedef __attribute__((address_space(3))) bf16x4* tn_lds_ptr;

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) char tn_smem[];
    auto As = [&](int buf) { return tn_smem + buf * (2 * kTnRows * kTnStride); };
    auto Bs = [&](int buf) { return tn_smem + buf * (2 * kTnRows * kTnStride) + kTnRows * kTnStride; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    // XCD-aware map (see gemm_nt_bf16_kernel): all output tiles of one token range read the same rows of A and B -- they
    // are placed on ONE XCD, so each row comes from HBM once rather than once per XCD.  Placement only.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, tiles = gx * gy;
        const int L = (bz * gy + by) * gx + bx, full = (gz / 8) * 8;
        const int xcd = L & 7, slot = L >> 3, z2 = (slot / tiles) * 8 + xcd, t2 = slot % tiles;
        if (L < full * tiles && z2 < full) { bz = z2; by = t2 / gx; bx = t2 - by * gx; }
    }
    const int n0 = by * 128, k0 = bx * 128;
    const int mbeg = bz * g.msplit_len, mend = min(g.M, mbeg + g.msplit_len);
    // staging role: 64 rows x 16 pieces (16 B) per operand tile, four per thread and operand
    const int srow = tid >> 4, spc = tid & 15;
    const bool a_ok = n0 + spc * 8 < g.N, b_ok = k0 + spc * 8 < g.K;
    uint4 ra[4], rb[4];
    auto load = [&](int m0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = m0 + srow + 16 * h;
            ra[h] = (a_ok && m < mend) ? *reinterpret_cast<const uint4*>(g.A + (size_t)m * g.lda + n0 + spc * 8) : uint4{0, 0, 0, 0};
            rb[h] = (b_ok && m < mend) ? *reinterpret_cast<const uint4*>(g.B + (size_t)m * g.ldb + k0 + spc * 8) : uint4{0, 0, 0, 0};
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = srow + 16 * h;
            *reinterpret_cast<uint4*>(As(buf) + row * kTnStride + spc * 16) = ra[h];
            *reinterpret_cast<uint4*>(Bs(buf) + row * kTnStride + spc * 16) = rb[h];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool cs_wave = g.colsum != nullptr && bx == 0 && wk == 0;          // (wave-uniform)
    f32x4 acs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    if (mbeg < mend) {
        load(mbeg);
        store(0);
        __syncthreads();
        int buf = 0;
        for (int m0 = mbeg; m0 < mend; m0 += kTnRows) {
            const bool more = m0 + kTnRows < mend;
            if (more) load(m0 + kTnRows);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // MFMA k-slot (q, j) <-> tile row 4q+j (j < 4) / 16+4q+(j-4) for both operands; lane r <-> column r of the tile
                const int row1 = kk * 32 + 4 * q + (r >> 2), row2 = row1 + 16, cl = 4 * (r & 3);
                bf16x8 an[4], bk[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ca = (wn * 64 + i * 16 + cl) * 2, cb = (wk * 64 + i * 16 + cl) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_ptr)(As(buf) + row1 * kTnStride + ca));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_ptr)(As(buf) + row2 * kTnStride + ca));
                    an[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_ptr)(Bs(buf) + row1 * kTnStride + cb));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tn_lds_ptr)(Bs(buf) + row2 * kTnStride + cb));
                    bk[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk[j], an[i], acc[i][j], 0, 0, 0);
                // the bias gradient rides along: the product with an all-ones B fragment is the column sum of the A tile in every
                // row of the result (exact products, fp32 accumulation) -- four more MFMAs per step in the first k-tile's
                // workgroups instead of a second pass over dY (colsum_T_kernel: 25 launches, 1.9 ms per transformer step)
                if (cs_wave) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, an[i], acs[i], 0, 0, 0);
                }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if (cs_wave && q == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + r;
            if (n < g.N) g.colsum[(size_t)bz * g.N + n] = acs[i][0];
        }
    }
    // lane (r, q) of tile (i, j): row n = i*16 + r, columns k = j*16 + 4q .. +3
    float* C = g.C + (size_t)bz * g.N * g.ldc;
    const bool vec4 = (g.K & 3) == 0 && (g.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + r;
        if (n >= g.N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kb = k0 + wk * 64 + j * 16 + 4 * q;
            if (kb >= g.K) continue;
            if (vec4) { *reinterpret_cast<f32x4*>(C + (size_t)n * g.ldc + kb) = acc[i][j]; continue; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kb + e < g.K) C[(size_t)n * g.ldc + kb + e] = acc[i][j][e];
        }
    }
}

// out[n][m] (bf16, leading dimension ldo >= M, columns M..ldo-1 zero) = in[m][n]; in is T with leading dimension ldi
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ in, uint16_t* __restrict__ out, int M, int N,
                                                            int ldi, int ldo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int m = m0 + ty + k, n = n0 + tx;
        tile[ty + k][tx] = (m < M && n < N) ? ldT<T>(in, (size_t)m * ldi + n) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int n = n0 + ty + k, m = m0 + tx;
        if (n < N && m < ldo) out[(size_t)n * ldo + m] = f2bf(tile[tx][ty + k]);
    }
}
// bf16 -> bf16, N % 8 == 0, ldi % 8 == 0, ldo % 8 == 0: 64 x 64 tiles, every global access a 16-byte piece (the weight-
// gradient GEMMs transpose two activation tensors per linear layer: the scalar form above cost 10 % of a step)
__global__ __launch_bounds__(256) void transpose_pad_bf16v_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                                  int M, int N, int ldi, int ldo) {
    __shared__ uint16_t tile[64][64 + 2];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (tid >> 3) + 32 * h, pc = tid & 7;            // row m, 8 columns n
        const int m = m0 + row, n = n0 + pc * 8;
        uint4 v = uint4{0, 0, 0, 0};
        if (m < M && n < N) v = *reinterpret_cast<const uint4*>(in + (size_t)m * ldi + n);
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[row][pc * 8 + k] = e[k];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = (tid >> 3) + 32 * h, pc = tid & 7;            // output row n, 8 columns m
        const int n = n0 + col, m = m0 + pc * 8;
        if (n < N && m < ldo) {
            uint4 v;
            uint16_t* e = reinterpret_cast<uint16_t*>(&v);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = tile[pc * 8 + k][col];
            *reinterpret_cast<uint4*>(out + (size_t)n * ldo + m) = v;
        }
    }
}

// bf16 copy with zero-padded rows: out[m][0:ldo] = in[m][0:N] | 0
template <typename T>
__global__ void cast_pad_kernel(const T* __restrict__ in, uint16_t* __restrict__ out, long long M, int N, int ldi, int ldo) {
    const size_t total = (size_t)M * ldo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / ldo; const int n = (int)(i - m * ldo);
        out[i] = n < N ? f2bf(ldT<T>(in, m * ldi + n)) : (uint16_t)0;
    }
}

// The bf16 weight cache of every linear layer in ONE launch: for each fp32 weight W (N, K) both copies the GEMMs read -- W as
// [N][ldo] (ldo >= K, pad columns zero) and W^T as [K][ldt] (ldt >= N, pad columns zero) -- from one read of W.  As 2 x 27 launches
// of cast_pad_kernel / transpose_pad_kernel (scalar accesses, W read twice: the policy layer alone is 933 MB per read) this was
// 1.15 ms of a 38 ms step.  table[j] = {W, out, outT, N, K, ldo, ldt, first tile of the job}; a workgroup owns a 64 x 64 tile:
// 16-byte pieces in, 16-byte pieces out both ways (the transpose through LDS).  Same rounding as the single kernels (f2bf).
__global__ __launch_bounds__(256) void weights16_multi_kernel(const long long* __restrict__ table, int njobs) {
    int lo = 0, hi = njobs - 1;                    // last job whose first tile is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)table[(size_t)mid * 8 + 7] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* t = table + (size_t)lo * 8;
    const float* __restrict__ W = reinterpret_cast<const float*>(t[0]);
    uint16_t* __restrict__ out = reinterpret_cast<uint16_t*>(t[1]);
    uint16_t* __restrict__ outT = reinterpret_cast<uint16_t*>(t[2]);
    const int N = (int)t[3], K = (int)t[4], ldo = (int)t[5], ldt = (int)t[6];
    const int ktiles = (ldo + 63) / 64, local = (int)blockIdx.x - (int)t[7];
    const int n0 = (local / ktiles) * 64, k0 = (local % ktiles) * 64;
    __shared__ uint16_t tile[64][64 + 2];
    const int tid = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (tid >> 3) + 32 * h, pc = tid & 7;            // row n, 8 columns k
        const int n = n0 + row, k = k0 + pc * 8;
        float f[8];
        if (n < N && k + 8 <= K && (K & 3) == 0) {            // (rows are 16-byte aligned only when K % 4 == 0)
            const f32x4 a = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k), b = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (n < N && k + e < K) ? W[(size_t)n * K + k + e] : 0.f;
        }
        uint4 v;
        uint16_t* e16 = reinterpret_cast<uint16_t*>(&v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { e16[e] = f2bf(f[e]); tile[row][pc * 8 + e] = e16[e]; }
        if (n < N && k < ldo) *reinterpret_cast<uint4*>(out + (size_t)n * ldo + k) = v;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = (tid >> 3) + 32 * h, pc = tid & 7;            // output row k, 8 columns n
        const int k = k0 + col, n = n0 + pc * 8;
        if (k < K && n < ldt) {
            uint4 v;
            uint16_t* e16 = reinterpret_cast<uint16_t*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) e16[e] = tile[pc * 8 + e][col];
            *reinterpret_cast<uint4*>(outT + (size_t)k * ldt + n) = v;
        }
    }
}

// ------------------------------------------------------------------ positional embedding
// x[b, s, :] += row_embed[s / 9] + col_embed[s % 9]     (transformer.py:84-87)
template <typename T>
__global__ void add_pos_kernel(T* __restrict__ x, const float* __restrict__ rowe, const float* __restrict__ cole, long long M, int d) {
    const size_t total = (size_t)M * d;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / d; const int c = (int)(i - m * d), s = (int)(m % 81);
        stT<T>(x, i, ldT<T>(x, i) + rowe[(s / 9) * d + c] + cole[(s % 9) * d + c]);
    }
}
// part[z][s, c] = sum over the z-th range of boards of dx[b, s, c] (fixed order); summed over z by sum_parts_kernel
template <typename T>
__global__ void pos_grad_kernel(const T* __restrict__ dx, float* __restrict__ part, int B, int d, int nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.y;
    if (i >= 81 * d) return;
    const int per = (B + nz - 1) / nz, lo = z * per, hi = min(B, lo + per);
    float s = 0.f;
    for (int b = lo; b < hi; ++b) s += ldT<T>(dx, (size_t)b * 81 * d + i);
    part[(size_t)z * 81 * d + i] = s;
}
// drow[r, c] = sum_col dpos[r*9+col, c]; dcol[col, c] = sum_r dpos[r*9+col, c]
__global__ void pos_grad_fold_kernel(const float* __restrict__ dpos, float* __restrict__ drow, float* __restrict__ dcol, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * d) return;
    const int a = i / d, c = i - a * d;
    float sr = 0.f, sc = 0.f;
    for (int k = 0; k < 9; ++k) { sr += dpos[(a * 9 + k) * d + c]; sc += dpos[(k * 9 + a) * d + c]; }
    drow[i] = sr; dcol[i] = sc;
}

// ------------------------------------------------------------------ layer norm (eps 1e-5, biased variance)
// one wave per row; mean / rstd kept in fp32 for the backward
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long long M, int d,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += ldT<T>(xr, c);
    const float mu = wave_sum(s) / d;
    float v = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = ldT<T>(xr, c) - mu; v += t * t; }
    const float rs = rsqrtf(wave_sum(v) / d + eps);
    for (int c = lane; c < d; c += 64) stT<T>(y + (size_t)row * d, c, (ldT<T>(xr, c) - mu) * rs * gamma[c] + beta[c]);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dx is ADDED to dres (the residual-stream gradient)
// when dres != NULL.  Per-workgroup partial sums of dgamma / dbeta: part[blockIdx.x][2][d].
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ part, long long M, int d,
                                                            int rows_per_block) {
    extern __shared__ float sm[];            // [4 waves][2][d] partial dgamma / dbeta (a lane always owns the same columns)
    for (int i = threadIdx.x; i < 8 * d; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = sm + wave * 2 * d;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (long long row = r0 + wave; row < min(M, r0 + rows_per_block); row += 4) {
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float g = ldT<T>(dy, (size_t)row * d + c) * gamma[c], xh = (ldT<T>(x, (size_t)row * d + c) - mu) * rs;
            s1 += g; s2 += g * xh;
        }
        s1 = wave_sum(s1) / d; s2 = wave_sum(s2) / d;
        for (int c = lane; c < d; c += 64) {
            const size_t o = (size_t)row * d + c;
            const float dyv = ldT<T>(dy, o), xh = (ldT<T>(x, o) - mu) * rs;
            float v = rs * (dyv * gamma[c] - s1 - xh * s2);
            if (dres) v += ldT<T>(dres, o);
            stT<T>(dx, o, v);
            mine[c] += dyv * xh;
            mine[d + c] += dyv;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += blockDim.x)
        part[(size_t)blockIdx.x * 2 * d + i] = ((sm[i] + sm[2 * d + i]) + sm[4 * d + i]) + sm[6 * d + i];
}
// 16-byte forms of the two kernels above for rows of L = d / (elements per 16 bytes) pieces, L a power of two <= 64
// (d = 256: half a wave per bf16 row).  A group of L lanes owns a row and keeps its piece in registers: one read of every
// operand, row sums by lane exchanges inside the group, and in the backward the dgamma / dbeta sums of a lane's columns
// stay in registers over all its rows (combined across the groups of a wave by lane exchanges, across the 4 waves through
// LDS in wave order: deterministic).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, T* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd, long long M,
                                                                int d, float eps, int L, int iters) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int P = E::kPer16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / L, pl = lane - sub * L, rpw = 64 / L;
    float gm[P], bt[P];
#pragma unroll
    for (int e = 0; e < P; ++e) { gm[e] = gamma[pl * P + e]; bt[e] = beta[pl * P + e]; }
    const float inv_d = 1.f / d;
    for (int it = 0; it < iters; ++it) {
        const long long row = (((long long)blockIdx.x * iters + it) * 4 + wave) * rpw + sub;
        const bool ok = row < M;
        float v[P];
        if (ok) E::unpack(*reinterpret_cast<const vec16*>(x + (size_t)row * d + pl * P), v);
        else {
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < P; ++e) s += v[e];
        for (int off = 1; off < L; off <<= 1) s += __shfl_xor(s, off);
        const float mu = s * inv_d;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < P; ++e) { v[e] -= mu; q2 += v[e] * v[e]; }
        for (int off = 1; off < L; off <<= 1) q2 += __shfl_xor(q2, off);
        const float rs = rsqrtf(q2 * inv_d + eps);
        if (ok) {
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = v[e] * rs * gm[e] + bt[e];
            *reinterpret_cast<vec16*>(y + (size_t)row * d + pl * P) = E::pack(v);
            if (pl == 0) { mean[row] = mu; rstd[row] = rs; }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const T* __restrict__ dres,
                                                                T* __restrict__ dx, float* __restrict__ part, long long M, int d,
                                                                int rows_per_block, int L, T* __restrict__ dx_drop, float drop_p,
                                                                unsigned long long seed) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int P = E::kPer16;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(drop_p);
    extern __shared__ float sm[];            // [4 waves][2][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / L, pl = lane - sub * L, rpw = 64 / L;
    float gm[P], dg[P], db[P];
#pragma unroll
    for (int e = 0; e < P; ++e) { gm[e] = gamma[pl * P + e]; dg[e] = 0.f; db[e] = 0.f; }
    const float inv_d = 1.f / d;
    const long long r0 = (long long)blockIdx.x * rows_per_block, rend = min(M, r0 + rows_per_block);
    // (every lane of a wave runs the same number of iterations: the lane exchanges need the whole wave)
    // Two row groups per iteration, everything either needs -- dy, x, the residual gradient, mean, rstd -- requested up front from
    // clamped row indices, then the two worked on in row order (same arithmetic, same order of the dgamma / dbeta additions).  As
    // a rolled loop with its loads behind `row < rend` and `dres != NULL` a row group was two dependent HBM round trips and nothing
    // of the next group was in flight: 3.1 TB/s on 3 passes.
    const T* drp = dres ? dres : dy;
    auto work = [&](long long row, bool ok, const vec16& q_dy, const vec16& q_x, const vec16& q_dr, float mu, float rs) __attribute__((always_inline)) {
        float g[P], xh[P], dyv[P];
        E::unpack(q_dy, dyv); E::unpack(q_x, xh);
        if (!ok) {
#pragma unroll
            for (int e = 0; e < P; ++e) { dyv[e] = 0.f; xh[e] = 0.f; }
            mu = 0.f; rs = 0.f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < P; ++e) {
            xh[e] = (xh[e] - mu) * rs;
            g[e] = dyv[e] * gm[e];
            s1 += g[e]; s2 += g[e] * xh[e];
        }
        for (int off = 1; off < L; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        s1 *= inv_d; s2 *= inv_d;
        if (ok) {
            float v[P];
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = rs * (g[e] - s1 - xh[e] * s2);
            if (dres) {
                float rr[P];
                E::unpack(q_dr, rr);
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] += rr[e];
            }
            const vec16 packed = E::pack(v);
            *reinterpret_cast<vec16*>(dx + (size_t)row * d + pl * P) = packed;
            if (dx_drop) {               // the same gradient through the dropout of the sub-layer below (its dY): mask of (seed, element)
                float w[P];
                E::unpack(packed, w);
#pragma unroll
                for (int e = 0; e < P; ++e) w[e] *= keep_scale(seed, (unsigned long long)row * d + pl * P + e, thresh, inv_keep);
                *reinterpret_cast<vec16*>(dx_drop + (size_t)row * d + pl * P) = E::pack(w);
            }
#pragma unroll
            for (int e = 0; e < P; ++e) { dg[e] += dyv[e] * xh[e]; db[e] += dyv[e]; }
        }
    };
    for (long long base = r0 + wave * rpw; base < rend; base += 8 * rpw) {
        const long long rowA = base + sub, rowB = base + 4 * rpw + sub;
        const bool okA = rowA < rend, okB = rowB < rend;
        const long long ra = min(rowA, rend - 1), rb = min(rowB, rend - 1);
        const size_t oa = (size_t)ra * d + pl * P, ob = (size_t)rb * d + pl * P;
        const vec16 dyA = *reinterpret_cast<const vec16*>(dy + oa), xA = *reinterpret_cast<const vec16*>(x + oa), drA = *reinterpret_cast<const vec16*>(drp + oa);
        const vec16 dyB = *reinterpret_cast<const vec16*>(dy + ob), xB = *reinterpret_cast<const vec16*>(x + ob), drB = *reinterpret_cast<const vec16*>(drp + ob);
        const float muA = mean[ra], rsA = rstd[ra], muB = mean[rb], rsB = rstd[rb];
        work(rowA, okA, dyA, xA, drA, muA, rsA);
        work(rowB, okB, dyB, xB, drB, muB, rsB);     // (base + 4 rpw >= rend for the whole wave: a pass of zeros, nothing stored)
    }
    for (int off = L; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < P; ++e) { dg[e] += __shfl_xor(dg[e], off); db[e] += __shfl_xor(db[e], off); }
    }
    if (lane < L) {
#pragma unroll
        for (int e = 0; e < P; ++e) { sm[wave * 2 * d + pl * P + e] = dg[e]; sm[wave * 2 * d + d + pl * P + e] = db[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += blockDim.x)
        part[(size_t)blockIdx.x * 2 * d + i] = ((sm[i] + sm[2 * d + i]) + sm[4 * d + i]) + sm[6 * d + i];
}
// out[i] = sum_p part[p][i]   (fixed order).  A 256-thread workgroup owns 32 columns: thread (column, g) sums the g-th
// eighth of the parts in order (column-contiguous 128-byte reads), thread (column, 0) then adds the eight partials in order.
// (One thread per column walked up to 2048 strided parts alone: 59 us for the 512 LayerNorm columns.)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int n) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    const int per = (nparts + 7) / 8, lo = g * per, hi = min(nparts, lo + per);
    float s = 0.f;
    if (i < n) {
        int p = lo;
        for (; p + 8 <= hi; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < hi; ++p) s += part[(size_t)p * n + i];
    }
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        out[i] = t;
    }
}

// ------------------------------------------------------------------ elementwise pieces of the backward
// out = in * keep(index) [* (act > 0)] [+ res]: dropout forward (fp32 mode, where the GEMM has no fused epilogue; + the
// residual stream) and dropout backward (optionally through the ReLU that precedes the dropout)
template <typename T>
__global__ void drop_apply_kernel(const T* __restrict__ g_in, const T* __restrict__ act, const T* __restrict__ res,
                                  T* __restrict__ g_out, long long n, float drop_p, unsigned long long seed, int vec) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int P = E::kPer16;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(drop_p);
    // 16-byte pieces when every pointer is 16-byte aligned (vec); the scalar loop covers n % P, or everything
    const size_t nv = vec ? (size_t)n / P : 0, gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    for (size_t v = gid; v < nv; v += gsz) {
        float x[P];
        E::unpack(*reinterpret_cast<const vec16*>(g_in + v * P), x);
        if (drop_p > 0.f) {
#pragma unroll
            for (int e = 0; e < P; ++e) x[e] *= keep_scale(seed, v * P + e, thresh, inv_keep);
        }
        if (act) {
            float a_[P];
            E::unpack(*reinterpret_cast<const vec16*>(act + v * P), a_);
#pragma unroll
            for (int e = 0; e < P; ++e) x[e] = a_[e] > 0.f ? x[e] : 0.f;
        }
        if (res) {
            float r_[P];
            E::unpack(*reinterpret_cast<const vec16*>(res + v * P), r_);
#pragma unroll
            for (int e = 0; e < P; ++e) x[e] += r_[e];
        }
        *reinterpret_cast<vec16*>(g_out + v * P) = E::pack(x);
    }
    for (size_t i = nv * P + gid; i < (size_t)n; i += gsz) {
        float v = ldT<T>(g_in, i);
        if (drop_p > 0.f) v *= keep_scale(seed, i, thresh, inv_keep);
        if (act && !(ldT<T>(act, i) > 0.f)) v = 0.f;
        if (res) v += ldT<T>(res, i);
        stT<T>(g_out, i, v);
    }
}
// column sums of a T matrix [M][N] in nsplit row ranges: part[s][n] (fp32), fixed order inside a range
template <typename T>
__global__ void colsum_T_kernel(const T* __restrict__ a, float* __restrict__ part, long long M, int N, int nsplit) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (n >= N) return;
    const long long per = (M + nsplit - 1) / nsplit, lo = s * per, hi = min(M, lo + per);
    float acc = 0.f;
    long long m = lo;
    for (; m + 8 <= hi; m += 8) {             // eight rows requested before the first is added (same order of additions)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ldT<T>(a, (size_t)(m + u) * N + n);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; m < hi; ++m) acc += ldT<T>(a, (size_t)m * N + n);
    part[(size_t)s * N + n] = acc;
}
// pooled[b, c] = mean_s x[b, s, c]; backward: dx[b, s, c] += dpooled[b, c] / 81 (+ dflat[b, s*d + c] when given)
template <typename T>
__global__ void mean_pool_kernel(const T* __restrict__ x, float* __restrict__ pooled, int B, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, c = i - b * d;
    float s = 0.f;
    for (int k = 0; k < 81; ++k) s += ldT<T>(x, ((size_t)b * 81 + k) * d + c);
    pooled[i] = s * (1.f / 81.f);
}
template <typename T>
__global__ void head_grad_kernel(const float* __restrict__ dpooled, const T* __restrict__ dflat, T* __restrict__ dx, long long M, int d) {
    const size_t total = (size_t)M * d;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / d; const int c = (int)(i - m * d);
        float v = dpooled ? dpooled[(m / 81) * d + c] * (1.f / 81.f) : 0.f;
        if (dflat) v += ldT<T>(dflat, i);
        stT<T>(dx, i, v);
    }
}
__global__ void tanh_kernel(float* __restrict__ v, long long n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * blockDim.x) v[i] = tanhf(v[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// ------------------------------------------------------------------ attention: one wave per (board, head)
// Operand convention of tile_mma: A[m][k] and B[n][k], both K-contiguous in LDS with element strides lda / ldb;
// acc tile (16 x 16): lane (r, q) holds rows 4q..4q+3 of column r.
template <typename T> struct Mm;
template <> struct Mm<bf16_t> {
    typedef uint16_t elem;
    static constexpr int kStep = 32;
    static __device__ __forceinline__ f32x4 run(const elem* A, int lda, const elem* B, int ldb, int kdim, int r, int q, f32x4 acc) {
        for (int k0 = 0; k0 < kdim; k0 += 32) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + r * lda + k0 + 8 * q);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + r * ldb + k0 + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        return acc;
    }
    static __device__ __forceinline__ elem cvt(float v) { return f2bf(v); }
    static __device__ __forceinline__ float up(elem v) { return bf2f(v); }
};
template <> struct Mm<float> {
    typedef float elem;
    static constexpr int kStep = 4;
    static __device__ __forceinline__ f32x4 run(const elem* A, int lda, const elem* B, int ldb, int kdim, int r, int q, f32x4 acc) {
        for (int k0 = 0; k0 < kdim; k0 += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * lda + k0 + q], B[r * ldb + k0 + q], acc, 0, 0, 0);
        return acc;
    }
    static __device__ __forceinline__ elem cvt(float v) { return v; }
    static __device__ __forceinline__ float up(elem v) { return v; }
};

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int kS = 81, kSP = 96;          // sequence, padded to 6 row tiles

struct AttnArgs {
    const void* qkv;          // [B*81][3d]: q | k | v, head h at columns h*dh
    void* out;                // [B*81][d]
    float* lse;               // [B][H][81]
    const void* dout;         // backward: [B*81][d]
    void* dqkv;               // backward: [B*81][3d]
    int B, H, dh, d;
    float scale, drop_p; unsigned long long seed;
};

// LDS sizes (elements) for head dimension dh: K-padded to the MFMA step, N-padded to 16
template <typename T> __host__ __device__ constexpr int attn_kp(int dh) { return (dh + Mm<T>::kStep - 1) / Mm<T>::kStep * Mm<T>::kStep; }
__host__ __device__ constexpr int attn_np(int dh) { return (dh + 15) / 16 * 16; }

template <typename T>
__global__ __launch_bounds__(64) void attention_fwd_kernel(AttnArgs a) {
    typedef typename Mm<T>::elem E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int dh = a.dh, KP = attn_kp<T>(dh), NP = attn_np(dh);
    const int ldq = KP + 8, ldv = kSP + 8, ldp = kSP + 8;
    E* Q = reinterpret_cast<E*>(smem);           // [96][ldq]   rows >= 81 and columns >= dh are zero
    E* K = Q + kSP * ldq;                        // [96][ldq]
    E* Vt = K + kSP * ldq;                       // [NP][ldv]   V transposed: Vt[c][s]
    E* P = Vt + NP * ldv;                        // [16][ldp]   one row tile of the (dropped) probabilities
    constexpr int VE = 16 / sizeof(E);       // elements per 16-byte piece (E and T have the same size)
    {
        uint4* z = reinterpret_cast<uint4*>(smem);
        const int n16 = (int)((size_t)(2 * kSP * ldq + NP * ldv + 16 * ldp) * sizeof(E) / 16);
        for (int i = lane; i < n16; i += 64) z[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();
    const T* base = static_cast<const T*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * dh;
    if (dh % VE == 0) {
        // 16-byte pieces: Q and K keep their layout (one 16-byte LDS store each), V is scattered into its transpose
        const int ppr = dh / VE;
        for (int i = lane; i < kS * ppr; i += 64) {
            const int s = i / ppr, c0 = (i - s * ppr) * VE;
            const T* row = base + (size_t)s * 3 * a.d + c0;
            const uint4 qv = *reinterpret_cast<const uint4*>(row);
            const uint4 kv = *reinterpret_cast<const uint4*>(row + a.d);
            const uint4 vv = *reinterpret_cast<const uint4*>(row + 2 * a.d);
            *reinterpret_cast<uint4*>(Q + s * ldq + c0) = qv;
            *reinterpret_cast<uint4*>(K + s * ldq + c0) = kv;
            const E* ve = reinterpret_cast<const E*>(&vv);
#pragma unroll
            for (int k = 0; k < VE; ++k) Vt[(c0 + k) * ldv + s] = ve[k];
        }
    } else {
        for (int i = lane; i < kS * dh; i += 64) {
            const int s = i / dh, c = i - s * dh;
            const T* row = base + (size_t)s * 3 * a.d;
            Q[s * ldq + c] = Mm<T>::cvt(ldT<T>(row, c));
            K[s * ldq + c] = Mm<T>::cvt(ldT<T>(row, a.d + c));
            Vt[c * ldv + s] = Mm<T>::cvt(ldT<T>(row, 2 * a.d + c));
        }
    }
    __syncthreads();
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p);
    for (int rt = 0; rt < 6; ++rt) {
        f32x4 sc[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            // acc^T trick: K rows as MFMA "A", Q rows as "B": lane (r, q) then holds score[row = rt*16 + r][col = ct*16 + 4q + i]
            sc[ct] = Mm<T>::run(K + ct * 16 * ldq, ldq, Q + rt * 16 * ldq, ldq, KP, r, q, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        // row of this lane: rt*16 + r; its 96 columns live in 6 tiles x 4 registers x the 4 lanes q that share r
        float mx = -INFINITY;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = ct * 16 + 4 * q + i;
                sc[ct][i] = col < kS ? sc[ct][i] * a.scale : -INFINITY;
                mx = fmaxf(mx, sc[ct][i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) { sc[ct][i] = __expf(sc[ct][i] - mx); sum += sc[ct][i]; }
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const int row = rt * 16 + r;
        const float inv = 1.f / sum;
        if (q == 0 && row < kS) a.lse[((size_t)b * a.H + h) * kS + row] = mx + __logf(sum);
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = ct * 16 + 4 * q + i;
                float p = sc[ct][i] * inv;
                if (a.drop_p > 0.f) p *= keep_scale(a.seed, ((unsigned long long)bh * kSP + row) * kSP + col, thresh, inv_keep);
                P[r * ldp + col] = Mm<T>::cvt(p);
            }
        __syncthreads();
        // O[row][c] = sum_col P[row][col] V[col][c]: A = Vt rows (c), B = P rows -> lane holds O[row = r][c = 4q + i]
        for (int nt = 0; nt < NP / 16; ++nt) {
            const f32x4 o = Mm<T>::run(Vt + nt * 16 * ldv, ldv, P, ldp, kSP, r, q, f32x4{0.f, 0.f, 0.f, 0.f});
            if (row < kS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = nt * 16 + 4 * q + i;
                    if (c < dh) stT<T>(static_cast<T*>(a.out), ((size_t)b * kS + row) * a.d + h * dh + c, o[i]);
                }
            }
        }
        __syncthreads();
    }
}

// backward of one (board, head): dQ, dK, dV from dO, with P recomputed from the saved log-sum-exp.
//   Pd = dropout(P);  dV = Pd^T dO;  dPd = dO V^T;  dP = dropout'(dPd);  dS = P * (dP - rowsum(dP * P)) * scale
//   dQ = dS K;  dK = dS^T Q
template <typename T, int NT>
__global__ __launch_bounds__(64) void attention_bwd_kernel(AttnArgs a) {
    typedef typename Mm<T>::elem E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int dh = a.dh, KP = attn_kp<T>(dh), NP = attn_np(dh);
    constexpr int RK = Mm<T>::kStep < 16 ? 16 : Mm<T>::kStep;   // rows per chunk = K extent of the transposed products
    const int ldq = KP + 8, ldt = kSP + 8, ldc = RK + 8;
    E* Q = reinterpret_cast<E*>(smem);        // natural [96][ldq]: Q, K, V, dO
    E* K = Q + kSP * ldq;
    E* V = K + kSP * ldq;
    E* dO = V + kSP * ldq;
    E* Qt = dO + kSP * ldq;                    // transposed [NP][ldt]: Qt[c][s], Kt, dOt
    E* Kt = Qt + NP * ldt;
    E* dOt = Kt + NP * ldt;
    E* dS = dOt + NP * ldt;                    // [RK][ldt]  rows of the chunk
    E* dSt = dS + RK * ldt;                    // [96][ldc]  the same tile transposed
    E* Pdt = dSt + kSP * ldc;                  // [96][ldc]  dropped probabilities, transposed
    const int total = 4 * kSP * ldq + 3 * NP * ldt + RK * ldt + 2 * kSP * ldc;
    constexpr int VE = 16 / sizeof(E);
    {
        uint4* z = reinterpret_cast<uint4*>(smem);
        const int n16 = (int)((size_t)total * sizeof(E) / 16);
        for (int i = lane; i < n16; i += 64) z[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();
    const T* base = static_cast<const T*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * dh;
    const T* dob = static_cast<const T*>(a.dout) + (size_t)b * kS * a.d + h * dh;
    if (dh % VE == 0) {
        const int ppr = dh / VE;
        for (int i = lane; i < kS * ppr; i += 64) {
            const int s = i / ppr, c0 = (i - s * ppr) * VE;
            const T* row = base + (size_t)s * 3 * a.d + c0;
            const uint4 qv = *reinterpret_cast<const uint4*>(row);
            const uint4 kv = *reinterpret_cast<const uint4*>(row + a.d);
            const uint4 vv = *reinterpret_cast<const uint4*>(row + 2 * a.d);
            const uint4 gv = *reinterpret_cast<const uint4*>(dob + (size_t)s * a.d + c0);
            *reinterpret_cast<uint4*>(Q + s * ldq + c0) = qv;
            *reinterpret_cast<uint4*>(K + s * ldq + c0) = kv;
            *reinterpret_cast<uint4*>(V + s * ldq + c0) = vv;
            *reinterpret_cast<uint4*>(dO + s * ldq + c0) = gv;
            const E* qe = reinterpret_cast<const E*>(&qv); const E* ke = reinterpret_cast<const E*>(&kv);
            const E* ge = reinterpret_cast<const E*>(&gv);
#pragma unroll
            for (int k = 0; k < VE; ++k) {
                Qt[(c0 + k) * ldt + s] = qe[k]; Kt[(c0 + k) * ldt + s] = ke[k]; dOt[(c0 + k) * ldt + s] = ge[k];
            }
        }
    } else {
        for (int i = lane; i < kS * dh; i += 64) {
            const int s = i / dh, c = i - s * dh;
            const T* row = base + (size_t)s * 3 * a.d;
            const float qv = ldT<T>(row, c), kv = ldT<T>(row, a.d + c), vv = ldT<T>(row, 2 * a.d + c), gv = ldT<T>(dob, (size_t)s * a.d + c);
            Q[s * ldq + c] = Mm<T>::cvt(qv); K[s * ldq + c] = Mm<T>::cvt(kv); V[s * ldq + c] = Mm<T>::cvt(vv); dO[s * ldq + c] = Mm<T>::cvt(gv);
            Qt[c * ldt + s] = Mm<T>::cvt(qv); Kt[c * ldt + s] = Mm<T>::cvt(kv); dOt[c * ldt + s] = Mm<T>::cvt(gv);
        }
    }
    __syncthreads();
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p);
    constexpr int NTMAX = NT;                  // 16-wide tiles of the head dimension (dh <= 64)
    f32x4 dK[6][NTMAX], dV[6][NTMAX];          // [column tile][dh tile]: lane holds dK[col = ct*16 + r][c = nt*16 + 4q + i]
#pragma unroll
    for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt) { dK[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    T* dq = static_cast<T*>(a.dqkv) + (size_t)b * kS * 3 * a.d + h * dh;
    for (int r0 = 0; r0 < kSP; r0 += RK) {
        for (int rt = 0; rt < RK / 16; ++rt) {
            const int row = r0 + rt * 16 + r;
            const float l = row < kS ? a.lse[((size_t)b * a.H + h) * kS + row] : 0.f;
            f32x4 p[6], dp[6];
            float D = 0.f;
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) {
                p[ct] = Mm<T>::run(K + ct * 16 * ldq, ldq, Q + (r0 + rt * 16) * ldq, ldq, KP, r, q, f32x4{0.f, 0.f, 0.f, 0.f});
                dp[ct] = Mm<T>::run(V + ct * 16 * ldq, ldq, dO + (r0 + rt * 16) * ldq, ldq, KP, r, q, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = ct * 16 + 4 * q + i;
                    const bool ok = col < kS && row < kS;
                    const float pv = ok ? __expf(p[ct][i] * a.scale - l) : 0.f;
                    float m = 1.f;
                    if (a.drop_p > 0.f) m = keep_scale(a.seed, ((unsigned long long)bh * kSP + row) * kSP + col, thresh, inv_keep);
                    const float dpv = ok ? dp[ct][i] * m : 0.f;        // gradient w.r.t. the un-dropped probability
                    p[ct][i] = pv; dp[ct][i] = dpv;
                    D += pv * dpv;
                    Pdt[col * ldc + rt * 16 + r] = Mm<T>::cvt(pv * m);
                }
            }
            D += __shfl_xor(D, 16); D += __shfl_xor(D, 32);
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = ct * 16 + 4 * q + i;
                    const E v = Mm<T>::cvt(p[ct][i] * (dp[ct][i] - D) * a.scale);
                    dS[(rt * 16 + r) * ldt + col] = v;
                    dSt[col * ldc + rt * 16 + r] = v;
                }
        }
        __syncthreads();
        // dQ rows of the chunk: dQ[row][c] = sum_col dS[row][col] K[col][c]: A = Kt rows (c), B = dS rows
        for (int rt = 0; rt < RK / 16; ++rt) {
            const int row = r0 + rt * 16 + r;
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 o = Mm<T>::run(Kt + nt * 16 * ldt, ldt, dS + rt * 16 * ldt, ldt, kSP, r, q, f32x4{0.f, 0.f, 0.f, 0.f});
                if (row < kS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = nt * 16 + 4 * q + i;
                        if (c < dh) stT<T>(dq, (size_t)row * 3 * a.d + c, o[i]);
                    }
                }
            }
        }
        // dK[col][c] += sum_row dS[row][col] Q[row][c]; dV[col][c] += sum_row Pd[row][col] dO[row][c]   (K extent = RK rows)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int nt = 0; nt < NTMAX; ++nt) {
                dK[ct][nt] = Mm<T>::run(Qt + nt * 16 * ldt + r0, ldt, dSt + ct * 16 * ldc, ldc, RK, r, q, dK[ct][nt]);
                dV[ct][nt] = Mm<T>::run(dOt + nt * 16 * ldt + r0, ldt, Pdt + ct * 16 * ldc, ldc, RK, r, q, dV[ct][nt]);
            }
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
        const int col = ct * 16 + r;
        if (col >= kS) continue;
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = nt * 16 + 4 * q + i;
                if (c < dh) {
                    stT<T>(dq, (size_t)col * 3 * a.d + a.d + c, dK[ct][nt][i]);
                    stT<T>(dq, (size_t)col * 3 * a.d + 2 * a.d + c, dV[ct][nt][i]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ bf16 attention with the operands in registers (dh <= 32)
// One wave per (board, head), four per workgroup.  Q, K, V, dO rows are loaded from global memory straight into MFMA
// fragments (16 bytes per lane: a row of the tile, 8 consecutive head dimensions).  The two orientations of a 16x16x32
// MFMA give a score tile either as [query = lane r][4 keys] (operands K, Q) or as [4 queries][key = lane r] (operands
// Q, K); in both, the four values a lane holds in two neighbouring tiles are exactly one operand fragment of the next
// product under the k-slot permutation {4q+j, 16+4q+j} -- the permutation the LDS transpose read delivers for the other
// operand.  So P (forward), dS and the dropped P (backward) never leave the registers; LDS holds one natural-layout
// [96][dh] matrix at a time, read once with ds_read_b64_tr_b16 into the transposed fragments (V^T; K^T, Q^T, dO^T), 9.6 KB
// per wave instead of 73 KB: 16 waves per CU instead of 2.  The backward computes each score / dP tile in both
// orientations (the MFMAs are not what this kernel waits for): pass A per query tile (softmax terms, D, dQ), pass B per
// key tile (dK, dV); D[query] travels between the passes through 96 floats of LDS.
constexpr int kAtStride = 96;                // bytes per row of the natural-layout LDS matrix (<= 64 B of data; 32 x odd)
constexpr int kAtWaveLds = kSP * kAtStride + 2 * kSP * 4;      // the matrix + two float rows (D, log-sum-exp) per wave
typedef __attribute__((address_space(3))) bf16x4* at_lds_ptr;

struct AtCtx {
    int lane, r, q, dh;
    char* buf;                               // this wave's [96][kAtStride] matrix
    // row-major fragment of row tile t, k-step ks, of a global matrix with row stride ldg (elements)
    __device__ __forceinline__ bf16x8 frag_g(const uint16_t* base, size_t ldg, int t, int ks) const {
        const int row = t * 16 + r, col = ks * 32 + q * 8;
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        return (row < kS && col < dh) ? *reinterpret_cast<const bf16x8*>(base + (size_t)row * ldg + col) : z;
    }
    // natural layout into LDS: rows >= 81 and columns >= dh are zero
    __device__ __forceinline__ void stage(const uint16_t* base, size_t ldg, int ppr) const {
        for (int i = lane; i < kSP * ppr; i += 64) {
            const int s = i / ppr, c0 = (i - s * ppr) * 8;
            uint4 v = {0, 0, 0, 0};
            if (s < kS && c0 < dh) v = *reinterpret_cast<const uint4*>(base + (size_t)s * ldg + c0);
            *reinterpret_cast<uint4*>(buf + s * kAtStride + c0 * 2) = v;
        }
    }
    // transposed fragment: column nt*16 + r of the LDS matrix at the rows (k-slots) pair*32 + {4q+j, 16+4q+j}
    __device__ __forceinline__ bf16x8 frag_t(int pair, int nt) const {
        const int row1 = pair * 32 + 4 * q + (r >> 2), cl = (nt * 16 + 4 * (r & 3)) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((at_lds_ptr)(buf + row1 * kAtStride + cl));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((at_lds_ptr)(buf + (row1 + 16) * kAtStride + cl));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
};
__device__ __forceinline__ bf16x8 at_pack(const f32x4& a, const f32x4& b) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (__bf16)a[e]; v[4 + e] = (__bf16)b[e]; }
    return v;
}

template <int NT>
__global__ __launch_bounds__(256) void attention_fwd_reg_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int wave = threadIdx.x >> 6;
    AtCtx c{(int)(threadIdx.x & 63), (int)(threadIdx.x & 15), (int)((threadIdx.x & 63) >> 4), a.dh, at_smem + wave * kAtWaveLds};
    const int r = c.r, q = c.q;
    const int bh = min(blockIdx.x * 4 + wave, a.B * a.H - 1);      // (a surplus wave repeats the last pair: same values, same stores)
    const int b = bh / a.H, h = bh - b * a.H;
    const uint16_t* qb = static_cast<const uint16_t*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const size_t ldg = 3 * (size_t)a.d;
    c.stage(qb + 2 * a.d, ldg, NT * 2);
    bf16x8 Kf[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) Kf[j] = c.frag_g(qb + a.d, ldg, j, 0);
    __syncthreads();
    bf16x8 Vt[3][NT];
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Vt[pr][nt] = c.frag_t(pr, nt);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p), dkey = drop_key(a.seed), ibase = (uint32_t)bh * kSP * kSP;
    uint16_t* ob = static_cast<uint16_t*>(a.out) + (size_t)b * kS * a.d + h * a.dh;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const bf16x8 Qf = c.frag_g(qb, ldg, i, 0);
        const int row = i * 16 + r;
        f32x4 sc[6];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[j], Qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[j][e] = (j * 16 + 4 * q + e < kS) ? sc[j][e] * a.scale : -INFINITY;
                mx = fmaxf(mx, sc[j][e]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[j][e] = __expf(sc[j][e] - mx); sum += sc[j][e]; }
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        if (q == 0 && row < kS) a.lse[((size_t)b * a.H + h) * kS + row] = mx + __logf(sum);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pv = sc[j][e] * inv;
                if (a.drop_p > 0.f) pv *= keep_scale32(dkey, ibase + row * kSP + j * 16 + 4 * q + e, thresh, inv_keep);
                sc[j][e] = pv;
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vt[pr][nt], at_pack(sc[2 * pr], sc[2 * pr + 1]), o, 0, 0, 0);
            const int cc = nt * 16 + 4 * q;        // lane holds O[row][cc .. cc+3]
            if (row < kS && cc < a.dh) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (__bf16)o[e];
                *reinterpret_cast<bf16x4*>(ob + (size_t)row * a.d + cc) = ov;
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attention_bwd_reg_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int wave = threadIdx.x >> 6;
    AtCtx c{(int)(threadIdx.x & 63), (int)(threadIdx.x & 15), (int)((threadIdx.x & 63) >> 4), a.dh, at_smem + wave * kAtWaveLds};
    float* Dq = reinterpret_cast<float*>(c.buf + kSP * kAtStride);          // [96]
    const int r = c.r, q = c.q;
    const int bh = min(blockIdx.x * 4 + wave, a.B * a.H - 1);
    const int b = bh / a.H, h = bh - b * a.H;
    const uint16_t* qb = static_cast<const uint16_t*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const uint16_t* gb = static_cast<const uint16_t*>(a.dout) + (size_t)b * kS * a.d + h * a.dh;
    uint16_t* dqb = static_cast<uint16_t*>(a.dqkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const float* lse = a.lse + ((size_t)b * a.H + h) * kS;
    const size_t ldg = 3 * (size_t)a.d;
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p), dkey = drop_key(a.seed), ibase = (uint32_t)bh * kSP * kSP;
    auto store4 = [&](uint16_t* base, int row, int cc, const f32x4& v) {    // 4 consecutive head dimensions of one token
        if (row < kS && cc < a.dh) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(base + (size_t)row * ldg + cc) = o;
        }
    };
    // ---- pass A: per query tile -- D[query] and dQ
    c.stage(qb + a.d, ldg, NT * 2);                                   // K
    {
        bf16x8 Kf[6];                                                  // (V fragments are fetched per tile: cache-resident)
#pragma unroll
        for (int j = 0; j < 6; ++j) Kf[j] = c.frag_g(qb + a.d, ldg, j, 0);
        __syncthreads();
        bf16x8 Kt[3][NT];
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Kt[pr][nt] = c.frag_t(pr, nt);
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {
            const bf16x8 Qf = c.frag_g(qb, ldg, i, 0), Gf = c.frag_g(gb, a.d, i, 0);
            const int row = i * 16 + r;
            const float l = row < kS ? lse[row] : 0.f;
            f32x4 p[6], dp[6];
            float D = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[j], Qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c.frag_g(qb + 2 * a.d, ldg, j, 0), Gf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = j * 16 + 4 * q + e;
                    const bool ok = col < kS && row < kS;
                    const float pv = ok ? __expf(p[j][e] * a.scale - l) : 0.f;
                    float m = 1.f;
                    if (a.drop_p > 0.f) m = keep_scale32(dkey, ibase + row * kSP + col, thresh, inv_keep);
                    const float dpv = ok ? dp[j][e] * m : 0.f;         // gradient w.r.t. the un-dropped probability
                    p[j][e] = pv; dp[j][e] = dpv;
                    D += pv * dpv;
                }
            }
            D += __shfl_xor(D, 16); D += __shfl_xor(D, 32);
            if (q == 0) Dq[row] = D;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) p[j][e] = p[j][e] * (dp[j][e] - D) * a.scale;      // dS
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
                    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kt[pr][nt], at_pack(p[2 * pr], p[2 * pr + 1]), o, 0, 0, 0);
                store4(dqb, row, nt * 16 + 4 * q, o);                   // dQ[row][nt*16 + 4q ..]
            }
        }
    }
    // ---- pass B: per key tile -- dK and dV (scores in the other orientation: lane = [4 queries][key r])
    __syncthreads();                                                  // K^T fragments are in registers: the matrix may go
    c.stage(qb, ldg, NT * 2);                                         // Q
    __syncthreads();
    bf16x8 Qt[3][NT], Gt[3][NT];                                      // (row-major Q / dO fragments are fetched per tile)
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Qt[pr][nt] = c.frag_t(pr, nt);
    __syncthreads();
    c.stage(gb, a.d, NT * 2);                                         // dO
    __syncthreads();
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Gt[pr][nt] = c.frag_t(pr, nt);
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        const bf16x8 Kf = c.frag_g(qb + a.d, ldg, j, 0), Vf = c.frag_g(qb + 2 * a.d, ldg, j, 0);
        const int col = j * 16 + r;
        f32x4 dK[NT], dV[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dK[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            f32x4 ds[2], pd[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int i = 2 * pr + hf;
                const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c.frag_g(qb, ldg, i, 0), Kf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c.frag_g(gb, a.d, i, 0), Vf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = i * 16 + 4 * q + e;
                    const bool ok = col < kS && row < kS;
                    const float pv = ok ? __expf(s[e] * a.scale - lse[min(row, kS - 1)]) : 0.f;
                    float m = 1.f;
                    if (a.drop_p > 0.f) m = keep_scale32(dkey, ibase + row * kSP + col, thresh, inv_keep);
                    const float dpv = ok ? g[e] * m : 0.f;
                    ds[hf][e] = pv * (dpv - Dq[row]) * a.scale;
                    pd[hf][e] = pv * m;
                }
            }
            const bf16x8 dsf = at_pack(ds[0], ds[1]), pdf = at_pack(pd[0], pd[1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                dK[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qt[pr][nt], dsf, dK[nt], 0, 0, 0);
                dV[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Gt[pr][nt], pdf, dV[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                              // lane holds d{K,V}[key = col][nt*16 + 4q ..]
            store4(dqb + a.d, col, nt * 16 + 4 * q, dK[nt]);
            store4(dqb + 2 * a.d, col, nt * 16 + 4 * q, dV[nt]);
        }
    }
}

// ------------------------------------------------------------------ the backward in two launches: dQ | dK, dV
// attention_bwd_reg_kernel spends its 1.23 ms (4096 boards, 8 heads of 32) waiting, not computing: 252 MFMAs per wave, three
// waves per SIMD (168 registers, 30 of them spilled), and pass B cannot start before pass A has D[query] = sum_key P dP.
// D is also rowsum(dO * O) with O the forward's (dropped) output -- the identity flash attention uses -- so with O handed in the
// two passes are independent kernels: twice the waves, each with the registers of ONE pass (no spills, 4-5 waves per SIMD), and a
// score tile of pass A is finished as soon as it is computed instead of waiting in registers for the row's D.
// D from the bf16 O differs from the fp32 sum by O's rounding (2^-9 relative): tests/test_hip_transformer.py bounds the result.
__device__ __forceinline__ void at_rowdot(const AtCtx& c, const uint16_t* gb, const uint16_t* ob, size_t ld, float* Dq) {
    for (int row = c.lane; row < kSP; row += 64) {
        float s = 0.f;
        if (row < kS) {
            for (int c0 = 0; c0 < c.dh; c0 += 8) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(gb + (size_t)row * ld + c0);
                const bf16x8 o = *reinterpret_cast<const bf16x8*>(ob + (size_t)row * ld + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf((float)g[e], (float)o[e], s);
            }
        }
        Dq[row] = s;
    }
}

#ifndef KA_ATT_DQ_OCC
#define KA_ATT_DQ_OCC 3
#endif
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KA_ATT_DQ_OCC))) void attention_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int wave = threadIdx.x >> 6;
    AtCtx c{(int)(threadIdx.x & 63), (int)(threadIdx.x & 15), (int)((threadIdx.x & 63) >> 4), a.dh, at_smem + wave * kAtWaveLds};
    float* Dq = reinterpret_cast<float*>(c.buf + kSP * kAtStride);          // [96]
    const int r = c.r, q = c.q;
    const int bh = min(blockIdx.x * 4 + wave, a.B * a.H - 1);
    const int b = bh / a.H, h = bh - b * a.H;
    const uint16_t* qb = static_cast<const uint16_t*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const uint16_t* gb = static_cast<const uint16_t*>(a.dout) + (size_t)b * kS * a.d + h * a.dh;
    const uint16_t* ob = static_cast<const uint16_t*>(a.out) + (size_t)b * kS * a.d + h * a.dh;
    uint16_t* dqb = static_cast<uint16_t*>(a.dqkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const float* lse = a.lse + ((size_t)b * a.H + h) * kS;
    const size_t ldg = 3 * (size_t)a.d;
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p), dkey = drop_key(a.seed), ibase = (uint32_t)bh * kSP * kSP;
    c.stage(qb + a.d, ldg, NT * 2);                                   // K
    at_rowdot(c, gb, ob, a.d, Dq);
    bf16x8 Kf[6], Vf[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { Kf[j] = c.frag_g(qb + a.d, ldg, j, 0); Vf[j] = c.frag_g(qb + 2 * a.d, ldg, j, 0); }
    __syncthreads();
    bf16x8 Kt[3][NT];
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Kt[pr][nt] = c.frag_t(pr, nt);
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const bf16x8 Qf = c.frag_g(qb, ldg, i, 0), Gf = c.frag_g(gb, a.d, i, 0);
        const int row = i * 16 + r;
        const float l = row < kS ? lse[row] : 0.f, D = Dq[row];
        f32x4 ds[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[j], Qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vf[j], Gf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = j * 16 + 4 * q + e;
                const bool ok = col < kS && row < kS;
                const float pv = ok ? __expf(p[e] * a.scale - l) : 0.f;
                float m = 1.f;
                if (a.drop_p > 0.f) m = keep_scale32(dkey, ibase + row * kSP + col, thresh, inv_keep);
                ds[j][e] = pv * ((ok ? dp[e] * m : 0.f) - D) * a.scale;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kt[pr][nt], at_pack(ds[2 * pr], ds[2 * pr + 1]), o, 0, 0, 0);
            const int cc = nt * 16 + 4 * q;
            if (row < kS && cc < a.dh) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (__bf16)o[e];
                *reinterpret_cast<bf16x4*>(dqb + (size_t)row * ldg + cc) = ov;
            }
        }
    }
}

#ifndef KA_ATT_DKV_OCC
#define KA_ATT_DKV_OCC 2      // two waves per SIMD with the query-side fragments in registers: 893 us for both passes against 1061 with three
#endif
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KA_ATT_DKV_OCC))) void attention_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int wave = threadIdx.x >> 6;
    AtCtx c{(int)(threadIdx.x & 63), (int)(threadIdx.x & 15), (int)((threadIdx.x & 63) >> 4), a.dh, at_smem + wave * kAtWaveLds};
    float* Dq = reinterpret_cast<float*>(c.buf + kSP * kAtStride);          // [96]
    const int r = c.r, q = c.q;
    const int bh = min(blockIdx.x * 4 + wave, a.B * a.H - 1);
    const int b = bh / a.H, h = bh - b * a.H;
    const uint16_t* qb = static_cast<const uint16_t*>(a.qkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const uint16_t* gb = static_cast<const uint16_t*>(a.dout) + (size_t)b * kS * a.d + h * a.dh;
    const uint16_t* ob = static_cast<const uint16_t*>(a.out) + (size_t)b * kS * a.d + h * a.dh;
    uint16_t* dqb = static_cast<uint16_t*>(a.dqkv) + (size_t)b * kS * 3 * a.d + h * a.dh;
    const float* lse = a.lse + ((size_t)b * a.H + h) * kS;
    const size_t ldg = 3 * (size_t)a.d;
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thresh = drop_thresh(a.drop_p), dkey = drop_key(a.seed), ibase = (uint32_t)bh * kSP * kSP;
    // D[query] and the queries' log-sum-exp through LDS: lane (key r, q) needs them for the queries 4q + e of every tile (read
    // from global memory per element, the 144 log-sum-exp loads per lane were what this pass waited for: 82 % of its wave
    // cycles in s_waitcnt, 238 vector-memory instructions per wave against 52 in the dQ pass)
    float* Ls = Dq + kSP;                                             // [96]
    at_rowdot(c, gb, ob, a.d, Dq);
    for (int row = c.lane; row < kSP; row += 64) Ls[row] = row < kS ? lse[row] : 0.f;
    c.stage(qb, ldg, NT * 2);                                         // Q
    __syncthreads();
    bf16x8 Qt[3][NT], Gt[3][NT];
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Qt[pr][nt] = c.frag_t(pr, nt);
    __syncthreads();
    c.stage(gb, a.d, NT * 2);                                         // dO
    __syncthreads();
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Gt[pr][nt] = c.frag_t(pr, nt);
    auto store4 = [&](uint16_t* base, int row, int cc, const f32x4& v) {
        if (row < kS && cc < a.dh) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(base + (size_t)row * ldg + cc) = o;
        }
    };
#if KA_ATT_DKV_OCC == 2
    bf16x8 Qr[6], Gr[6];                                               // row-major query / dO fragments: the same for every key tile
#pragma unroll
    for (int i = 0; i < 6; ++i) { Qr[i] = c.frag_g(qb, ldg, i, 0); Gr[i] = c.frag_g(gb, a.d, i, 0); }
#define KA_QR(i_) Qr[i_]
#define KA_GR(i_) Gr[i_]
#else
#define KA_QR(i_) c.frag_g(qb, ldg, i_, 0)
#define KA_GR(i_) c.frag_g(gb, a.d, i_, 0)
#endif
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        const bf16x8 Kf = c.frag_g(qb + a.d, ldg, j, 0), Vf = c.frag_g(qb + 2 * a.d, ldg, j, 0);
        const int col = j * 16 + r;
        f32x4 dK[NT], dV[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dK[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            f32x4 ds[2], pd[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int i = 2 * pr + hf;
                const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KA_QR(i), Kf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KA_GR(i), Vf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = i * 16 + 4 * q + e;
                    const bool ok = col < kS && row < kS;
                    const float pv = ok ? __expf(sc[e] * a.scale - Ls[row]) : 0.f;
                    float m = 1.f;
                    if (a.drop_p > 0.f) m = keep_scale32(dkey, ibase + row * kSP + col, thresh, inv_keep);
                    const float dpv = ok ? g[e] * m : 0.f;
                    ds[hf][e] = pv * (dpv - Dq[row]) * a.scale;
                    pd[hf][e] = pv * m;
                }
            }
            const bf16x8 dsf = at_pack(ds[0], ds[1]), pdf = at_pack(pd[0], pd[1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                dK[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qt[pr][nt], dsf, dK[nt], 0, 0, 0);
                dV[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Gt[pr][nt], pdf, dV[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            store4(dqb + a.d, col, nt * 16 + 4 * q, dK[nt]);
            store4(dqb + 2 * a.d, col, nt * 16 + 4 * q, dV[nt]);
        }
    }
}
#undef KA_QR
#undef KA_GR

template <typename T> size_t attn_fwd_lds(int dh) {
    const int KP = attn_kp<T>(dh), NP = attn_np(dh);
    return sizeof(typename Mm<T>::elem) * (size_t)(2 * kSP * (KP + 8) + NP * (kSP + 8) + 16 * (kSP + 8));
}
template <typename T> size_t attn_bwd_lds(int dh) {
    const int KP = attn_kp<T>(dh), NP = attn_np(dh);
    const int RK = Mm<T>::kStep < 16 ? 16 : Mm<T>::kStep;
    return sizeof(typename Mm<T>::elem) * (size_t)(4 * kSP * (KP + 8) + 3 * NP * (kSP + 8) + RK * (kSP + 8) + 2 * kSP * (RK + 8));
}

// lanes per row of the 16-byte LayerNorm kernels (0: use the one-wave-per-row forms)
inline int ln_vec_lanes(int d, int dtype, uintptr_t ptr_bits) {
    const int P = dtype == KA_DTYPE_BF16 ? 8 : 4;
    if (d % P != 0 || (ptr_bits & 15)) return 0;
    const int L = d / P;
    return (L >= 1 && L <= 64 && (L & (L - 1)) == 0) ? L : 0;
}
inline int grid1d(size_t n, int cap) { const size_t b = (n + 255) / 256; return (int)(b < (size_t)cap ? (b ? b : 1) : cap); }

}  // namespace

#define KA_TF_DISPATCH(dtype, stmt)                                                                  \
    do {                                                                                             \
        if ((dtype) == KA_DTYPE_BF16) { typedef bf16_t T; stmt; }                                    \
        else if ((dtype) == KA_DTYPE_F32) { typedef float T; stmt; }                                 \
        else { ka_set_error("transformer: unknown dtype %d", (dtype)); return KA_ERR_ARG; }          \
    } while (0)

// ------------------------------------------------------------------ C ABI
// C[M][ldc] = epilogue(A[M][lda] * B[N][ldb]^T): bf16 operands (K % 32 == 0, 16-byte aligned rows), fp32 accumulation.
// nsplit > 1: C receives nsplit fp32 slabs [nsplit][M][ldc] of partial sums over K ranges (no epilogue; reduce with
// ka_reduce_slabs).  Replaces nn.Linear / its input- and weight-gradient GEMMs of transformer.py:40-61 under autocast.
static int tf_gemm_nt_impl(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* relu_act,
                           int M, int N, int K, int lda, int ldb, int ldc, int c_bf16, int relu, int nsplit, float drop_p,
                           unsigned long long seed, void* stream);
extern "C" int ka_tf_gemm_nt(const void* A, const void* B, void* C, const float* bias, const void* residual, int M, int N, int K,
                             int lda, int ldb, int ldc, int c_bf16, int relu, int nsplit, float drop_p, unsigned long long seed,
                             void* stream) {
    return tf_gemm_nt_impl(A, B, C, bias, residual, nullptr, M, N, K, lda, ldb, ldc, c_bf16, relu, nsplit, drop_p, seed, stream);
}
// the input-gradient GEMM of a layer that is followed by ReLU + dropout in the forward (FFN linear1): the dropout mask of
// (seed, element) and the ReLU mask of the saved activation are applied in the epilogue -- C = (A B^T) * keep * [relu_act > 0]
extern "C" int ka_tf_gemm_nt_masked(const void* A, const void* B, void* C, const void* relu_act, int M, int N, int K, int lda,
                                    int ldb, int ldc, float drop_p, unsigned long long seed, void* stream) {
    KA_REQUIRE(relu_act, "tf_gemm_nt_masked: null activation");
    return tf_gemm_nt_impl(A, B, C, nullptr, nullptr, relu_act, M, N, K, lda, ldb, ldc, 1, 0, 1, drop_p, seed, stream);
}
static int tf_gemm_nt_impl(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* relu_act,
                           int M, int N, int K, int lda, int ldb, int ldc, int c_bf16, int relu, int nsplit, float drop_p,
                           unsigned long long seed, void* stream) {
    KA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "tf_gemm_nt: bad arguments");
    KA_REQUIRE(K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0, "tf_gemm_nt: K %% 32 and lda/ldb %% 8 required (K=%d lda=%d ldb=%d)", K, lda, ldb);
    KA_REQUIRE(nsplit >= 1 && (nsplit == 1 || (!bias && !residual && !relu && !c_bf16 && drop_p == 0.f)), "tf_gemm_nt: split-K slabs carry no epilogue");
    int len = K;
    if (nsplit > 1) { len = ((K + kBK - 1) / kBK + nsplit - 1) / nsplit * kBK; nsplit = (K + len - 1) / len; }
    NtArgs g{static_cast<const uint16_t*>(A), static_cast<const uint16_t*>(B), C, bias, residual, relu_act, M, N, K, lda, ldb, ldc,
             c_bf16, relu, len, 1, drop_p, seed};
    g.lds_epilogue = ka_opt(KA_OPT_TF_LDS_EPI, g.lds_epilogue);
    // K = 256 (d_model of BASELINE config 5), bf16 output, whole 16-column runs: the activation-stationary form (KA_TF_K256=0: off)
    const bool k256 = ka_opt(KA_OPT_TF_K256, 1) != 0;
    if (k256 && K == kKsK && nsplit == 1 && c_bf16 && N % kKsCols == 0 && N <= 4096 && ldc % 8 == 0 && lda >= K && ldb >= K) {
        const size_t lds = kKsLds + (size_t)N * sizeof(float);
        const int full = M / kKsRows, tail = M % kKsRows ? 1 : 0;
        hipStream_t st = static_cast<hipStream_t>(stream);
#define KA_K256(FULL_, RES_, ACT_, GRID_, ARGS_)                                                                              \
        {                                                                                                                     \
            static std::atomic<unsigned long long> d{0};                                                                      \
            if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&gemm_nt_k256_kernel<FULL_, RES_, ACT_>), d, "tf_gemm_nt (k256)")) return rc; \
            hipLaunchKernelGGL((gemm_nt_k256_kernel<FULL_, RES_, ACT_>), dim3(GRID_), dim3(256), lds, st, ARGS_);             \
        }
#define KA_K256_FORMS(FULL_, GRID_, ARGS_)                                                                                    \
        if (residual && relu_act) KA_K256(FULL_, true, true, GRID_, ARGS_)                                                    \
        else if (residual) KA_K256(FULL_, true, false, GRID_, ARGS_)                                                          \
        else if (relu_act) KA_K256(FULL_, false, true, GRID_, ARGS_)                                                          \
        else KA_K256(FULL_, false, false, GRID_, ARGS_)
        if (full > 0) { KA_K256_FORMS(true, full, g) }
        if (tail) {                                              // the ragged last panel: one more workgroup of the guarded instantiation
            NtArgs t = g;
            t.row0 = full * kKsRows;
            KA_K256_FORMS(false, 1, t)
        }
#undef KA_K256_FORMS
#undef KA_K256
        return ka_check_launch("tf_gemm_nt (k256)");
    }
    static std::atomic<unsigned long long> done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&gemm_nt_bf16_kernel), done, "tf_gemm_nt")) return rc;
    // all three extents large, nothing but a bias in the epilogue: the 256 x 256 tile form (KA_TF_BIG=0: off)
    if (ka_opt(KA_OPT_TF_BIG, 1) != 0 && nsplit == 1 && M >= 1024 && N >= 1024 && K >= 1024 && K % kBK == 0 && !residual && !relu_act &&
        !relu && drop_p == 0.f && (size_t)M * lda < (1ull << 31) && (size_t)N * ldb < (1ull << 31)) {      // (32-bit operand byte offsets)
        g.map_gx = (N + kGN - 1) / kGN; g.map_gy = (M + kGM - 1) / kGM;
        const int supers = ((g.map_gx + 7) / 8) * ((g.map_gy + 3) / 4);
        static std::atomic<unsigned long long> dbig{0};
        if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&gemm_nt_big_kernel), dbig, "tf_gemm_nt (big)")) return rc;
        hipLaunchKernelGGL(gemm_nt_big_kernel, dim3(((supers + 7) / 8) * 8 * 32), dim3(512), kGLds, static_cast<hipStream_t>(stream), g);
        return ka_check_launch("tf_gemm_nt (big)");
    }
    const int gx = (N + kBN - 1) / kBN, gy = (M + kBM - 1) / kBM;
    dim3 grid(gx, gy, nsplit);
    if (gx > 8 && gy >= 8 && ka_opt(KA_OPT_TF_MAP2D, 1) != 0) {      // (0: the one-m-tile-per-XCD map for every shape)
        g.map_gx = gx; g.map_gy = gy;
        const int supers = ((gx + 7) / 8) * ((gy + 7) / 8);
        grid = dim3(((supers + 7) / 8) * 8 * 64, 1, nsplit);
    }
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, grid, dim3(256), 2 * (kBM + kBN) * kLdsStride, static_cast<hipStream_t>(stream), g);
    return ka_check_launch("tf_gemm_nt");
}
// C[z][N][ldc] (fp32 slabs, z < ka_tf_gemm_tn_slabs(M, nsplit)) = partial sums over token ranges of A^T B, A [M][lda] and
// B [M][ldb] row-major bf16 with N resp. K columns (N, K, lda, ldb multiples of 8: 16-byte pieces).  One slab: C is the
// result.  Replaces autograd's weight gradient of nn.Linear (transformer.py:40-61) without transposed operand copies.
extern "C" int ka_tf_gemm_tn_slabs(int M, int nsplit) {
    if (nsplit <= 1) return 1;
    const int len = ((M + kTnRows - 1) / kTnRows + nsplit - 1) / nsplit * kTnRows;
    return (M + len - 1) / len;
}
static int tf_gemm_tn_impl(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, int lda, int ldb, int ldc,
                           int nsplit, void* stream);
extern "C" int ka_tf_gemm_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int nsplit,
                             void* stream) {
    return tf_gemm_tn_impl(A, B, C, nullptr, M, N, K, lda, ldb, ldc, nsplit, stream);
}
// ... and the layer's bias gradient in the same launch: colsum [ka_tf_gemm_tn_slabs(M, nsplit)][N] = per-slab column sums of A
// (db = their sum; transformer.py:40-61, the bias of nn.Linear)
extern "C" int ka_tf_gemm_tn_bias(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, int lda, int ldb,
                                  int ldc, int nsplit, void* stream) {
    KA_REQUIRE(colsum, "tf_gemm_tn_bias: null column-sum slab");
    return tf_gemm_tn_impl(A, B, C, colsum, M, N, K, lda, ldb, ldc, nsplit, stream);
}
static int tf_gemm_tn_impl(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, int lda, int ldb, int ldc,
                           int nsplit, void* stream) {
    KA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nsplit >= 1, "tf_gemm_tn: bad arguments");
    KA_REQUIRE(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= N && ldb >= K && ldc >= K,
               "tf_gemm_tn: N, K, lda, ldb must be multiples of 8 (N=%d K=%d lda=%d ldb=%d ldc=%d)", N, K, lda, ldb, ldc);
    const int ns = ka_tf_gemm_tn_slabs(M, nsplit);
    const int len = ns == 1 ? M : ((M + kTnRows - 1) / kTnRows + nsplit - 1) / nsplit * kTnRows;
    TnArgs g{static_cast<const uint16_t*>(A), static_cast<const uint16_t*>(B), C, M, N, K, lda, ldb, ldc, len, colsum};
    static std::atomic<unsigned long long> done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&gemm_tn_bf16_kernel), done, "tf_gemm_tn")) return rc;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((K + 127) / 128, (N + 127) / 128, ns), dim3(256), 4 * kTnRows * kTnStride,
                       static_cast<hipStream_t>(stream), g);
    return ka_check_launch("tf_gemm_tn");
}
// number of slabs ka_tf_gemm_nt writes for a requested split (the K ranges are whole 64-steps)
extern "C" int ka_tf_gemm_nt_slabs(int K, int nsplit) {
    if (nsplit <= 1) return 1;
    const int len = ((K + kBK - 1) / kBK + nsplit - 1) / nsplit * kBK;
    return (K + len - 1) / len;
}

// out[n][m] = bf16(in[m][n]), rows zero-padded to ldo (>= M, a multiple of 32): the K-contiguous operand form of the
// weight-gradient GEMM (contraction over tokens) and of the transposed weight cache
extern "C" int ka_tf_transpose_pad(const void* in, void* out, int M, int N, int ldi, int ldo, int dtype, void* stream) {
    KA_REQUIRE(in && out && ldo >= M, "tf_transpose_pad: bad arguments");
    if (dtype == KA_DTYPE_BF16 && N % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0) {
        hipLaunchKernelGGL(transpose_pad_bf16v_kernel, dim3((N + 63) / 64, (ldo + 63) / 64), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(in), static_cast<uint16_t*>(out), M, N,
                           ldi, ldo);
        return ka_check_launch("tf_transpose_pad");
    }
    dim3 grid((N + 31) / 32, (ldo + 31) / 32);
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(transpose_pad_kernel<T>, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                                             static_cast<const T*>(in), static_cast<uint16_t*>(out), M, N, ldi, ldo));
    return ka_check_launch("tf_transpose_pad");
}
// the bf16 weight cache of n linear layers in one launch (weights16_multi_kernel): table rows {W fp32 (N, K) row-major, out (N, ldo)
// bf16, outT (K, ldt) bf16, N, K, ldo, ldt, first tile}; K % 4 == 0, ldo % 8 == 0, ldt % 8 == 0, every pointer 16-byte aligned;
// total_tiles = sum over the jobs of ceil(ldt / 64) * ceil(ldo / 64)
extern "C" int ka_tf_weights16_multi(const void* table, int n, int total_tiles, void* stream) {
    KA_REQUIRE(table && n > 0 && total_tiles > 0, "tf_weights16_multi: bad arguments");
    hipLaunchKernelGGL(weights16_multi_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const long long*>(table), n);
    return ka_check_launch("tf_weights16_multi");
}
extern "C" int ka_tf_cast_pad(const void* in, void* out, long long M, int N, int ldi, int ldo, int dtype, void* stream) {
    KA_REQUIRE(in && out && ldo >= N, "tf_cast_pad: bad arguments");
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(cast_pad_kernel<T>, dim3(grid1d((size_t)M * ldo, 4096)), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), static_cast<const T*>(in),
                                             static_cast<uint16_t*>(out), M, N, ldi, ldo));
    return ka_check_launch("tf_cast_pad");
}

extern "C" int ka_tf_add_pos(void* x, const float* row_embed, const float* col_embed, int B, int d, int dtype, void* stream) {
    KA_REQUIRE(x && row_embed && col_embed, "tf_add_pos: null tensor");
    const long long M = (long long)B * 81;
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(add_pos_kernel<T>, dim3(grid1d((size_t)M * d, 4096)), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), static_cast<T*>(x), row_embed, col_embed, M, d));
    return ka_check_launch("tf_add_pos");
}
// scratch: 65 * 81 * d floats
extern "C" int ka_tf_pos_grad(const void* dx, float* scratch, float* drow, float* dcol, int B, int d, int dtype, void* stream) {
    KA_REQUIRE(dx && scratch && drow && dcol, "tf_pos_grad: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nz = B < 64 ? B : 64, n = 81 * d;
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(pos_grad_kernel<T>, dim3((n + 255) / 256, nz), dim3(256), 0, st,
                                             static_cast<const T*>(dx), scratch + n, B, d, nz));
    hipLaunchKernelGGL(sum_parts_kernel, dim3((n + 31) / 32), dim3(256), 0, st, scratch + n, scratch, nz, n);
    hipLaunchKernelGGL(pos_grad_fold_kernel, dim3((9 * d + 255) / 256), dim3(256), 0, st, scratch, drow, dcol, d);
    return ka_check_launch("tf_pos_grad");
}

extern "C" int ka_tf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   long long M, int d, float eps, int dtype, void* stream) {
    KA_REQUIRE(x && gamma && beta && y && mean && rstd, "tf_layernorm_fwd: null tensor");
    if (const int L = ln_vec_lanes(d, dtype, reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y))) {
        const int rpw = 64 / L, iters = 4;
        const long long rows_per_block = 4LL * rpw * iters;
        KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_fwd_vec_kernel<T>, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)),
                                                 dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const T*>(x), gamma, beta,
                                                 static_cast<T*>(y), mean, rstd, M, d, eps, L, iters));
        return ka_check_launch("tf_layernorm_fwd");
    }
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), static_cast<const T*>(x), gamma, beta,
                                             static_cast<T*>(y), mean, rstd, M, d, eps));
    return ka_check_launch("tf_layernorm_fwd");
}
extern "C" int ka_tf_layernorm_parts(long long M) { const long long p = (M + 127) / 128; return (int)(p < 2048 ? p : 2048); }
// dx = LayerNorm'(dy) [+ dres]; dgamma / dbeta [d] via part (ka_tf_layernorm_parts(M) * 2 * d floats)
extern "C" int ka_tf_drop_apply(const void* g_in, const void* act, const void* res, void* g_out, long long n, float drop_p,
                                unsigned long long seed, int dtype, void* stream);
// sums of the [dgamma | dbeta] part rows straight into the two gradient tensors
// (1024 threads = 32 row groups per 32 columns, four rows in flight per thread: with 8 groups the 2048 part rows of a B = 4096 step
//  were a 256-deep chain of dependent row reads per thread -- 100 us for 4 MB, 12 launches per step; fixed group order: deterministic)
__global__ __launch_bounds__(1024) void sum_parts2_kernel(const float* __restrict__ part, float* __restrict__ outa, float* __restrict__ outb,
                                                          int nparts, int d) {
    __shared__ float red[32][32];
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5, n = 2 * d;
    const int i = blockIdx.x * 32 + col;
    const int per = (nparts + 31) / 32, lo = g * per, hi = min(nparts, lo + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = lo;
        for (; p + 3 < hi; p += 4) {
            s0 += part[(size_t)p * n + i]; s1 += part[(size_t)(p + 1) * n + i];
            s2 += part[(size_t)(p + 2) * n + i]; s3 += part[(size_t)(p + 3) * n + i];
        }
        for (; p < hi; ++p) s0 += part[(size_t)p * n + i];
    }
    red[g][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 32; ++u) t += red[u][col];
        if (i < d) outa[i] = t; else outb[i - d] = t;
    }
}
extern "C" int ka_tf_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                        const void* dres, void* dx, void* dx_drop, float drop_p, unsigned long long seed, float* part,
                                        float* dgamma, float* dbeta, long long M, int d, int dtype, void* stream);
extern "C" int ka_tf_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                   const void* dres, void* dx, float* part, float* dgamma, float* dbeta, long long M, int d,
                                   int dtype, void* stream) {
    return ka_tf_layernorm_bwd_drop(dy, x, gamma, mean, rstd, dres, dx, nullptr, 0.f, 0, part, dgamma, dbeta, M, d, dtype, stream);
}
// ... and, when dx_drop != NULL, a second output dx_drop = dx * dropout_keep(seed, element): the gradient entering the
// sub-layer below through its dropout (the elementwise pass this saves read and wrote the token tensor once more)
extern "C" int ka_tf_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                        const void* dres, void* dx, void* dx_drop, float drop_p, unsigned long long seed, float* part,
                                        float* dgamma, float* dbeta, long long M, int d, int dtype, void* stream) {
    KA_REQUIRE(dy && x && gamma && mean && rstd && dx && part && dgamma && dbeta, "tf_layernorm_bwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nparts = ka_tf_layernorm_parts(M);
    const int rpb = (int)((M + nparts - 1) / nparts);
    if (const int L = ln_vec_lanes(d, dtype, reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) |
                                              reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(dx))) {
        KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_bwd_vec_kernel<T>, dim3(nparts), dim3(256), 8 * d * sizeof(float), st,
                                                 static_cast<const T*>(dy), static_cast<const T*>(x), gamma, mean, rstd,
                                                 static_cast<const T*>(dres), static_cast<T*>(dx), part, M, d, rpb, L,
                                                 static_cast<T*>(dx_drop), drop_p, seed));
    } else {
        KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3(nparts), dim3(256), 8 * d * sizeof(float), st,
                                                 static_cast<const T*>(dy), static_cast<const T*>(x), gamma, mean, rstd,
                                                 static_cast<const T*>(dres), static_cast<T*>(dx), part, M, d, rpb));
        if (dx_drop) {               // (rows the 16-byte kernel does not cover: the mask as its own pass)
            if (int rc = ka_tf_drop_apply(dx, nullptr, nullptr, dx_drop, M * d, drop_p, seed, dtype, stream)) return rc;
        }
    }
    // part rows are [dgamma | dbeta]
    hipLaunchKernelGGL(sum_parts2_kernel, dim3((2 * d + 31) / 32), dim3(1024), 0, st, part, dgamma, dbeta, nparts, d);
    return ka_check_launch("tf_layernorm_bwd");
}

extern "C" int ka_tf_drop_apply(const void* g_in, const void* act, const void* res, void* g_out, long long n, float drop_p,
                                unsigned long long seed, int dtype, void* stream) {
    KA_REQUIRE(g_in && g_out && n > 0, "tf_drop_apply: bad arguments");
    const uintptr_t bits = reinterpret_cast<uintptr_t>(g_in) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(res) |
                           reinterpret_cast<uintptr_t>(g_out);
    const int vec = (bits & 15) == 0;
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(drop_apply_kernel<T>, dim3(grid1d((size_t)n / 4, 4096)), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), static_cast<const T*>(g_in),
                                             static_cast<const T*>(act), static_cast<const T*>(res), static_cast<T*>(g_out),
                                             n, drop_p, seed, vec));
    return ka_check_launch("tf_drop_apply");
}
// bias gradient: out[n] = sum_m a[m][n]; part: nsplit * N floats
extern "C" int ka_tf_colsum(const void* a, float* part, float* out, long long M, int N, int nsplit, int dtype, void* stream) {
    KA_REQUIRE(a && part && out && nsplit >= 1, "tf_colsum: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(colsum_T_kernel<T>, dim3((N + 255) / 256, nsplit), dim3(256), 0, st,
                                             static_cast<const T*>(a), part, M, N, nsplit));
    hipLaunchKernelGGL(sum_parts_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part, out, nsplit, N);
    return ka_check_launch("tf_colsum");
}
extern "C" int ka_tf_mean_pool(const void* x, float* pooled, int B, int d, int dtype, void* stream) {
    KA_REQUIRE(x && pooled, "tf_mean_pool: null tensor");
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(mean_pool_kernel<T>, dim3((B * d + 255) / 256), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), static_cast<const T*>(x), pooled, B, d));
    return ka_check_launch("tf_mean_pool");
}
// dx[b,s,:] = dpooled[b,:] / 81 + dflat[b, s, :]   (gradient of the two heads w.r.t. the encoder output)
extern "C" int ka_tf_head_grad(const float* dpooled, const void* dflat, void* dx, int B, int d, int dtype, void* stream) {
    KA_REQUIRE(dx && (dpooled || dflat), "tf_head_grad: null tensor");
    const long long M = (long long)B * 81;
    KA_TF_DISPATCH(dtype, hipLaunchKernelGGL(head_grad_kernel<T>, dim3(grid1d((size_t)M * d, 4096)), dim3(256), 0,
                                             static_cast<hipStream_t>(stream), dpooled, static_cast<const T*>(dflat),
                                             static_cast<T*>(dx), M, d));
    return ka_check_launch("tf_head_grad");
}
extern "C" int ka_tf_tanh(float* v, long long n, void* stream) {
    KA_REQUIRE(v && n > 0, "tf_tanh: bad arguments");
    hipLaunchKernelGGL(tanh_kernel, dim3(grid1d((size_t)n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), v, n);
    return ka_check_launch("tf_tanh");
}
extern "C" int ka_tf_tanh_bwd(const float* dy, const float* y, float* dx, long long n, void* stream) {
    KA_REQUIRE(dy && y && dx && n > 0, "tf_tanh_bwd: bad arguments");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid1d((size_t)n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, dx, n);
    return ka_check_launch("tf_tanh_bwd");
}

// Multi-head self-attention over the 81 squares (nn.MultiheadAttention inside nn.TransformerEncoderLayer,
// transformer.py:45-55): out = softmax(Q K^T / sqrt(dh)) V per (board, head), with dropout on the probabilities in
// training; lse [B][H][81] is kept for the backward.  dh <= 64.
extern "C" int ka_tf_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int dh, float drop_p,
                                   unsigned long long seed, int dtype, void* stream) {
    KA_REQUIRE(qkv && out && lse && B > 0 && H > 0 && dh > 0 && dh <= 64, "tf_attention_fwd: bad arguments (dh <= 64)");
    AttnArgs a{qkv, out, lse, nullptr, nullptr, B, H, dh, H * dh, 1.0f / sqrtf((float)dh), drop_p, seed};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool reg_form = dtype == KA_DTYPE_BF16 && dh <= 32 && dh % 8 == 0 && (H * dh) % 8 == 0 && (long long)B * H * kSP * kSP < (1LL << 32) &&
                          !ka_opt_set(KA_OPT_TF_ATTN_LDS);
    if (reg_form) {           // operands in registers, four (board, head) pairs per workgroup
        const int grid = (B * H + 3) / 4;
        if (dh <= 16) hipLaunchKernelGGL(attention_fwd_reg_kernel<1>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        else          hipLaunchKernelGGL(attention_fwd_reg_kernel<2>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        return ka_check_launch("tf_attention_fwd");
    }
    if (dtype == KA_DTYPE_BF16) {
        static std::atomic<unsigned long long> done{0};
        if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&attention_fwd_kernel<bf16_t>), done, "tf_attention_fwd")) return rc;
        hipLaunchKernelGGL(attention_fwd_kernel<bf16_t>, dim3(B * H), dim3(64), attn_fwd_lds<bf16_t>(dh), st, a);
    } else if (dtype == KA_DTYPE_F32) {
        static std::atomic<unsigned long long> done{0};
        if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&attention_fwd_kernel<float>), done, "tf_attention_fwd")) return rc;
        hipLaunchKernelGGL(attention_fwd_kernel<float>, dim3(B * H), dim3(64), attn_fwd_lds<float>(dh), st, a);
    } else { ka_set_error("tf_attention_fwd: unknown dtype %d", dtype); return KA_ERR_ARG; }
    return ka_check_launch("tf_attention_fwd");
}
// The backward with the forward's output handed in (D = rowsum(dO * O)): dQ and dK / dV as two independent launches on the
// register-resident path (bf16, dh <= 32); every other shape / dtype goes to ka_tf_attention_bwd and ignores `out`.
extern "C" int ka_tf_attention_bwd(const void* qkv, const void* dout, const float* lse, void* dqkv, int B, int H, int dh,
                                   float drop_p, unsigned long long seed, int dtype, void* stream);
extern "C" int ka_tf_attention_bwd_o(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int H,
                                     int dh, float drop_p, unsigned long long seed, int dtype, void* stream) {
    KA_REQUIRE(qkv && out && dout && lse && dqkv && B > 0 && H > 0 && dh > 0 && dh <= 64, "tf_attention_bwd_o: bad arguments (dh <= 64)");
    if (dtype == KA_DTYPE_BF16 && dh <= 32 && dh % 8 == 0 && (H * dh) % 8 == 0 && (long long)B * H * kSP * kSP < (1LL << 32) &&
        !ka_opt_set(KA_OPT_TF_ATTN_LDS) && !ka_opt_set(KA_OPT_TF_ATTN_ONE)) {
        AttnArgs a{qkv, const_cast<void*>(out), const_cast<float*>(lse), dout, dqkv, B, H, dh, H * dh, 1.0f / sqrtf((float)dh), drop_p, seed};
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int grid = (B * H + 3) / 4;
        if (dh <= 16) {
            hipLaunchKernelGGL(attention_bwd_dq_kernel<1>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
            hipLaunchKernelGGL(attention_bwd_dkv_kernel<1>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        } else {
            hipLaunchKernelGGL(attention_bwd_dq_kernel<2>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
            hipLaunchKernelGGL(attention_bwd_dkv_kernel<2>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        }
        return ka_check_launch("tf_attention_bwd_o");
    }
    return ka_tf_attention_bwd(qkv, dout, lse, dqkv, B, H, dh, drop_p, seed, dtype, stream);
}
extern "C" int ka_tf_attention_bwd(const void* qkv, const void* dout, const float* lse, void* dqkv, int B, int H, int dh,
                                   float drop_p, unsigned long long seed, int dtype, void* stream) {
    KA_REQUIRE(qkv && dout && lse && dqkv && B > 0 && H > 0 && dh > 0 && dh <= 64, "tf_attention_bwd: bad arguments (dh <= 64)");
    AttnArgs a{qkv, nullptr, const_cast<float*>(lse), dout, dqkv, B, H, dh, H * dh, 1.0f / sqrtf((float)dh), drop_p, seed};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = attn_np(dh) / 16;
    if (dtype == KA_DTYPE_BF16 && dh <= 32 && dh % 8 == 0 && (H * dh) % 8 == 0 && (long long)B * H * kSP * kSP < (1LL << 32) &&
        !ka_opt_set(KA_OPT_TF_ATTN_LDS)) {
        const int grid = (B * H + 3) / 4;
        if (dh <= 16) hipLaunchKernelGGL(attention_bwd_reg_kernel<1>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        else          hipLaunchKernelGGL(attention_bwd_reg_kernel<2>, dim3(grid), dim3(256), 4 * kAtWaveLds, st, a);
        return ka_check_launch("tf_attention_bwd");
    }
#define KA_ATTN_BWD(T_, NT_)                                                                                              \
    do {                                                                                                                  \
        static std::atomic<unsigned long long> done{0};                                                                   \
        if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&attention_bwd_kernel<T_, NT_>), done, "tf_attention_bwd")) return rc; \
        hipLaunchKernelGGL((attention_bwd_kernel<T_, NT_>), dim3(B * H), dim3(64), attn_bwd_lds<T_>(dh), st, a);           \
    } while (0)
    if (dtype == KA_DTYPE_BF16) {
        if (nt == 1) KA_ATTN_BWD(bf16_t, 1); else if (nt == 2) KA_ATTN_BWD(bf16_t, 2); else if (nt == 3) KA_ATTN_BWD(bf16_t, 3); else KA_ATTN_BWD(bf16_t, 4);
    } else if (dtype == KA_DTYPE_F32) {
        if (nt == 1) KA_ATTN_BWD(float, 1); else if (nt == 2) KA_ATTN_BWD(float, 2); else if (nt == 3) KA_ATTN_BWD(float, 3); else KA_ATTN_BWD(float, 4);
    } else { ka_set_error("tf_attention_bwd: unknown dtype %d", dtype); return KA_ERR_ARG; }
#undef KA_ATTN_BWD
    return ka_check_launch("tf_attention_bwd");
}
