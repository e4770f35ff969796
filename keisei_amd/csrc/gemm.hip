// Small dense contractions of the SE-ResNet: global-pool FC, squeeze-excite FC, 1x1 policy
// convolutions and the value / score heads (forward, input-gradient and weight-gradient forms).
//
//   C[M,N] = act( opA(A)[M,K] * opB(B)[K,N] + bias[N] )
//
// These are < 0.4 % of the model's FLOPs; they run on the exact-f32 matrix
// instruction (64x64x32 LDS tiles) with split-K for the weight-gradient forms whose reduction runs
// over B*81 rows.  Operands may be fp32 or bf16 (activations) and are widened on load, rows need no
// alignment (the 139-wide logit rows); accumulation is always fp32.
//
// Replaces nn.Linear / 1x1 nn.Conv2d at se_resnet.py:57-61,65-66,120-130 and their backward.
#include "common.h"

namespace {

struct GemmArgs {
    const void* A; const void* B; void* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int transA, transB;          // opA(A)[m,k] = transA ? A[k*lda+m] : A[m*lda+k]; same for B[k,n]
    int a_bf16, b_bf16, c_bf16, relu, ksplit_len, accumulate;
    int a_vec, b_vec;            // rows of the operand are 16-byte (fp32) / 8-byte (bf16) aligned: 4-element vector loads
};

// (operand pointers may come from a device table -- the grouped weight-gradient launch -- where the compiler sees generic
// pointers and would emit FLAT loads, which also count on the LDS counter; every access goes through global-space views)
#define KA_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float ldx(const void* p, size_t i, int bf16) {
    return bf16 ? bf2f(((const KA_GLOBAL uint16_t*)p)[i]) : ((const KA_GLOBAL float*)p)[i];
}

// four consecutive elements of an fp32 / bf16 operand starting at element i, widened to fp32; `valid` (<= 0 .. >= 4) of
// them exist, the rest read as zero.  One vector load when the operand's rows are aligned, element loads otherwise
// (rows of 139 logits, ragged K).
__device__ __forceinline__ f32x4 ld4(const void* base, size_t i, int bf16, int vec, int valid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid <= 0) return v;
    if (vec && valid >= 4) {
        if (bf16) {
            const uint32_t ux = ((const KA_GLOBAL uint32_t*)((const KA_GLOBAL uint16_t*)base + i))[0];
            const uint32_t uy = ((const KA_GLOBAL uint32_t*)((const KA_GLOBAL uint16_t*)base + i))[1];
            const uint2 u = {ux, uy};
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        } else {
            v = *(const KA_GLOBAL f32x4*)((const KA_GLOBAL float*)base + i);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < valid) v[e] = ldx(base, i + e, bf16);
    }
    return v;
}

// fp32 or bf16 operands (widened on load, so every product is exact in fp32).  Four-element global loads
// along the operand's contiguous axis, next K-tile prefetched into registers while the current one is multiplied
// on the matrix cores with the exact-f32 v_mfma_f32_16x16x4_f32 (wave w owns rows 16w..16w+15 of the 64x64 tile and
// all four 16-column tiles: 32 MFMAs per K-tile and wave, one ds_read_b32 per operand fragment, row stride 80 floats
// = conflict-free for the (row/col, k) lane layout).
// NW = waves per workgroup: 4 (each wave a 16x64 strip) or 16 (each wave ONE 16x16 tile) -- the latter for problems with
// few workgroups (small-batch rollout inference), where the chain of 32 dependent-issue f32 MFMAs per K-tile and wave,
// not the machine, sets the time; only the first 256 threads stage the tiles.
// ONES (grouped weight-gradient form): B carries a virtual extra column of ones at n == N, so that output column is the
// column sum of A over K -- the bias gradient -- and goes to `colsum_out` instead of C.
template <bool TA, bool TB, int NW, bool ONES>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int bx, int by, int bz, float* colsum_out) {
    constexpr int BK = 32;
    __shared__ __attribute__((aligned(16))) float As[BK][64 + 16];
    __shared__ __attribute__((aligned(16))) float Bs[BK][64 + 16];
    const int tid = threadIdx.x;
    const int m0 = by * 64, n0 = bx * 64;
    const int kbeg = bz * g.ksplit_len, kend = min(g.K, kbeg + g.ksplit_len);
    const int Neff = g.N + ((ONES && colsum_out) ? 1 : 0);
    // per-thread load roles: two float4 per operand per tile
    //   contiguous-K operand (A not transposed / B transposed): row = t>>3 (+32), k4 = t&7
    //   contiguous-M/N operand: k = t>>4 (+16), col4 = t&15
    f32x4 ra[2], rb[2];
    const bool loader = NW == 4 || tid < 256;
    auto load_tile = [&](int k0) {
        if (!loader) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!TA) {
                const int m = min(m0 + (tid >> 3) + 32 * h, g.M - 1), k = k0 + (tid & 7) * 4;
                ra[h] = ld4(g.A, (size_t)m * g.lda + k, g.a_bf16, g.a_vec, kend - k);
            } else {
                const int k = k0 + (tid >> 4) + 16 * h, m = m0 + (tid & 15) * 4;
                ra[h] = ld4(g.A, (size_t)k * g.lda + m, g.a_bf16, g.a_vec, k < kend ? g.M - m : 0);
            }
            if (TB) {
                const int n = min(n0 + (tid >> 3) + 32 * h, g.N - 1), k = k0 + (tid & 7) * 4;
                rb[h] = ld4(g.B, (size_t)n * g.ldb + k, g.b_bf16, g.b_vec, kend - k);
            } else {
                const int k = k0 + (tid >> 4) + 16 * h, n = n0 + (tid & 15) * 4;
                rb[h] = ld4(g.B, (size_t)k * g.ldb + n, g.b_bf16, g.b_vec, k < kend ? g.N - n : 0);
                if (ONES && Neff > g.N && k < kend && g.N >= n && g.N < n + 4) rb[h][g.N - n] = 1.f;
            }
        }
    };
    auto store_tile = [&]() {
        if (!loader) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!TA) {
                const int m = (tid >> 3) + 32 * h, k = (tid & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) As[k + e][m] = ra[h][e];
            } else {
                *reinterpret_cast<f32x4*>(&As[(tid >> 4) + 16 * h][(tid & 15) * 4]) = ra[h];
            }
            if (TB) {
                const int n = (tid >> 3) + 32 * h, k = (tid & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[k + e][n] = rb[h][e];
            } else {
                *reinterpret_cast<f32x4*>(&Bs[(tid >> 4) + 16 * h][(tid & 15) * 4]) = rb[h];
            }
        }
    };
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;      // 16-row strip; for NW == 16 also the wave's 16-column tile
    constexpr int NJ = NW == 4 ? 4 : 1;           // column tiles per wave
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool rows_live = m0 + wm * 16 < g.M;
    load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (k0 + BK < kend) load_tile(k0 + BK);          // in flight during the MFMA block
        if (!rows_live) continue;                        // strip beyond M (skinny weight-gradient forms): barriers only
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nt = (NW == 4 ? j : wn) * 16;
            if (n0 + nt >= Neff) continue;               // uniform: column tiles beyond N (N = 32 heads) cost nothing
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4)           // A[m = 16 wm + r][k = kk + q] x B[k][n = nt + r]
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[kk + q][wm * 16 + r], Bs[kk + q][nt + r], acc[j], 0, 0, 0);
        }
    }
    // accumulator lane (r, q), element i: C[m = 16 wm + 4q + i][n = 16 j + r]
    const size_t slab = (size_t)bz * g.M * g.ldc;
    KA_GLOBAL float* C = (KA_GLOBAL float*)g.C;
    if (!rows_live) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + (NW == 4 ? j : wn) * 16 + r;
        if (n >= Neff) continue;
        if (ONES && n == g.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 16 + 4 * q + i;
                if (m < g.M) ((KA_GLOBAL float*)colsum_out)[m] = acc[j][i];
            }
            continue;
        }
        const float bv = g.bias ? ((const KA_GLOBAL float*)g.bias)[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 16 + 4 * q + i;
            if (m >= g.M) continue;
            float v = acc[j][i] + bv;
            if (g.relu) v = fmaxf(v, 0.f);
            const size_t o = slab + (size_t)m * g.ldc + n;
            if (g.c_bf16) ((KA_GLOBAL uint16_t*)g.C)[o] = f2bf(v);
            else C[o] = g.accumulate ? C[o] + v : v;
        }
    }
}

template <bool TA, bool TB, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_fast_kernel(GemmArgs g) {
    gemm_tile<TA, TB, NW, false>(g, blockIdx.x, blockIdx.y, blockIdx.z, nullptr);
}

// All FC weight / bias gradients of a backward pass in ONE launch (168 jobs at 40 blocks):
//   job j: dW_j (N,K) = dY_j^T X_j,  db_j (N) = column sums of dY_j      (dY_j (M,N) fp32, X_j (M rows of ldx, fp32 or bf16))
// table[j] = {dY, X, dW, db (or 0), M, N, K, ldx, x_bf16, first workgroup of the job}; a workgroup owns one 64x64 tile of one
// job and runs the whole contraction over the M rows (no split-K: one fixed summation order, nothing to reduce).  As single
// launches these were 6 GEMM + 8 slab-reduce + 4 column-sum launches per block on the critical stream of the backward.
__global__ __launch_bounds__(256) void gemm_grouped_wgrad_kernel(const long long* __restrict__ table, int njobs) {
    int lo = 0, hi = njobs - 1;                    // last job whose first workgroup is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)table[(size_t)mid * 10 + 9] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* t = table + (size_t)lo * 10;
    const int M = (int)t[4], N = (int)t[5], K = (int)t[6], ldx = (int)t[7];
    float* db = reinterpret_cast<float*>(t[3]);
    GemmArgs g{reinterpret_cast<const void*>(t[0]), reinterpret_cast<const void*>(t[1]), reinterpret_cast<void*>(t[2]), nullptr,
               N, K, M, N, ldx, K, 1, 0, 0, (int)t[8], 0, 0, (M + 31) / 32 * 32, 0, 0, 0};
    g.a_vec = (N % 4 == 0) && ((t[0] & 15) == 0);
    g.b_vec = (ldx % 4 == 0) && ((t[1] & (g.b_bf16 ? 7 : 15)) == 0);
    const int tiles_x = (K + (db ? 1 : 0) + 63) / 64;
    const int local = (int)blockIdx.x - (int)t[9];
    gemm_tile<true, false, 4, true>(g, local % tiles_x, local / tiles_x, 0, db);
}

// Two chained small FC layers of the board-vector paths (global-pool bias: 3C -> G -> C; squeeze-excite: C -> C/r -> 2C;
// value / score heads) in ONE launch:
//     y = W2 * relu(W1 * x' + b1) + b2,      x' = x                      (in_scale == NULL)
//                                            x' = in_scale[k] * (x * in_alpha) + in_shift[k]   (the SE squeeze from the
//                                                                         conv's per-board sums: BN affine and 1/81)
// A 512-thread workgroup owns 16 rows.  x' is staged once in LDS; phase 1 spreads the H/16 hidden column tiles (and, when
// there are fewer than 8 of them, K ranges) over the 8 waves, exact-f32 MFMAs with the weight rows read straight from
// L2 as 16-byte pieces (lane (r, q) holds k = k0+4q .. +3 of row r: MFMA i contracts k0+4q+i on both operands); the
// ReLU'd hidden rows go through LDS; phase 2 spreads the N2/16 output tiles over the waves.  The launches this replaces
// (an affine kernel and two 64x64-tile GEMMs) are latency-bound at every batch size of this model: 2 x 17 us at 128
// rows (rollout inference), 2 x 25-60 us at 4096.
struct ChainArgs {
    const float* x; const float* in_scale; const float* in_shift; float in_alpha;
    const float* W1; const float* b1; const float* W2; const float* b2;
    float* x_out;        // optional (M,K1): x' (kept for the backward)
    float* hidden_out;   // optional (M,H): relu(W1 x' + b1)
    float* y;            // (M,N2)
    int M, K1, ldx, H, N2;
    const float* hmask;  // optional (M,H): hidden = (W1 x') * [hmask > 0], no bias, no ReLU -- the backward of a chain (ReLU mask of the saved hidden)
    unsigned long long* stamps;   // diagnostic only (ka_debug_conv_stamps + a -DKA_DIAG_FC_TL build, tools/_diag/fc_chain_tl.py): per-wave phase stamps
};

constexpr int kFcBatch = 12;     // weight pieces of a wave in flight together (48 registers)
__global__ __launch_bounds__(512) void fc_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xs_ld = a.K1 + 4, hs_ld = a.H + 4;
    float* xs = lds;                          // [16][K1+4]
    float* hs = xs + 16 * xs_ld;              // [16][H+4]
    float* part = hs + 16 * hs_ld;            // [ksplit][16][H]   (only when H/16 < 8)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16;
#ifdef KA_DIAG_FC_TL
#define KA_FTL(ph) do { if (a.stamps && blockIdx.x % 100 == 0 && blockIdx.x < 300 && lane == 0) \
        a.stamps[((blockIdx.x / 100) * 8 + wave) * 8 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KA_FTL(ph) do {} while (0)
#endif
    KA_FTL(0);
    float touch = 0.f;       // (kept alive to the kernel's last statement, so that nothing waits for these reads on their own)
    // ---- the phase-2 weights start their way into the L2 now (one 4-byte read per 64-byte piece, spread over the workgroup,
    // results discarded): phase 2 then finds them there instead of paying a fabric trip per tile
    {
        const size_t w2_pieces = ((size_t)a.N2 * a.H * sizeof(float) + 63) / 64;
        for (size_t i = tid; i < w2_pieces; i += 512) touch += a.W2[i * 16];
    }
    // ---- stage x' (rows beyond M repeat the last row; their results are never stored)
    const int k4n = a.K1 >> 2;
    for (int i = tid; i < 16 * k4n; i += 512) {
        const int row = i / k4n, c4 = (i - row * k4n) * 4;
        const int m = min(m0 + row, a.M - 1);
        f32x4 v = *reinterpret_cast<const f32x4*>(a.x + (size_t)m * a.ldx + c4);
        if (a.in_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.in_scale + c4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.in_shift + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sc[e] * (v[e] * a.in_alpha) + sh[e];
            if (a.x_out && m0 + row < a.M) *reinterpret_cast<f32x4*>(a.x_out + (size_t)m * a.K1 + c4) = v;
        }
        *reinterpret_cast<f32x4*>(xs + row * xs_ld + c4) = v;
    }
    KA_FTL(1);
    __syncthreads();
    KA_FTL(2);
    // ---- phase 1: hidden = relu(x' W1^T + b1)
    const int T1 = a.H >> 4;
    const int ksplit = T1 >= 8 ? 1 : 8 / T1;            // T1 in {1,2,4} -> 8,4,2 K ranges per tile
    {
        const int klen = a.K1 / ksplit;
        for (int unit = wave; unit < T1 * ksplit; unit += 8) {
            const int tile = unit % T1, kp = unit / T1;
            const float* wrow = a.W1 + (size_t)(tile * 16 + r) * a.K1 + 4 * q;
            const float* xrow = xs + r * xs_ld + 4 * q;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int kbeg = kp * klen, kend = kbeg + klen;
            // The weights of a layer are NOT in the L2 when its chain kernel starts (a step's worth of 170 MB tensors has passed
            // since; tools/_diag/fc_chain_cold.py times the kernel that way).  The pieces go out kFcBatch at a time, then the MFMAs
            // run on them in k order (same products, same order: bit-identical): 4 trips instead of 12 for the 768-wide chain --
            // cold, with the touch above, 38 -> 32 us; the per-wave stamps (tools/_diag/fc_chain_tl.py) still show phase 1 at ~27 k
            // cycles against 9 k of exact-f32 matrix work, with one or four accumulators per tile alike (NOTES_r04 section 17).
            for (int k0 = kbeg; k0 < kend; k0 += 16 * kFcBatch) {
                f32x4 bv[kFcBatch];
#pragma unroll
                for (int u = 0; u < kFcBatch; ++u)
                    if (k0 + 16 * u < kend) bv[u] = *reinterpret_cast<const f32x4*>(wrow + k0 + 16 * u);
#pragma unroll
                for (int u = 0; u < kFcBatch; ++u) {
                    if (k0 + 16 * u >= kend) break;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(xrow + k0 + 16 * u);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[u][i], acc, 0, 0, 0);
                }
            }
            // acc lane (r, q), element i: hidden[m = 4q+i][n = 16 tile + r]
            if (ksplit == 1) {
                const float bias = a.b1 ? a.b1[tile * 16 + r] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float hv = fmaxf(acc[i] + bias, 0.f);
                    if (a.hmask) hv = a.hmask[(size_t)min(m0 + 4 * q + i, a.M - 1) * a.H + tile * 16 + r] > 0.f ? acc[i] : 0.f;
                    hs[(4 * q + i) * hs_ld + tile * 16 + r] = hv;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) part[(kp * 16 + 4 * q + i) * a.H + tile * 16 + r] = acc[i];
            }
        }
    }
    KA_FTL(3);
    __syncthreads();
    KA_FTL(4);
    if (ksplit > 1) {
        for (int i = tid; i < 16 * a.H; i += 512) {
            const int row = i / a.H, n = i - row * a.H;
            float v = a.b1 ? a.b1[n] : 0.f;
            for (int kp = 0; kp < ksplit; ++kp) v += part[(kp * 16 + row) * a.H + n];
            if (a.hmask) hs[row * hs_ld + n] = a.hmask[(size_t)min(m0 + row, a.M - 1) * a.H + n] > 0.f ? v : 0.f;
            else hs[row * hs_ld + n] = fmaxf(v, 0.f);
        }
        __syncthreads();
    }
    if (a.hidden_out)
        for (int i = tid; i < 16 * a.H; i += 512) {
            const int row = i / a.H, n = i - row * a.H;
            if (m0 + row < a.M) a.hidden_out[(size_t)(m0 + row) * a.H + n] = hs[row * hs_ld + n];
        }
    KA_FTL(5);
    // ---- phase 2: y = hidden W2^T + b2
    const int T2 = (a.N2 + 15) >> 4;
    for (int tile = wave; tile < T2; tile += 8) {
        const int n = min(tile * 16 + r, a.N2 - 1);
        const float* wrow = a.W2 + (size_t)n * a.H + 4 * q;
        const float* hrow = hs + r * hs_ld + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < a.H; k0 += 16 * kFcBatch) {      // (as phase 1: the tile's weight pieces in one trip)
            f32x4 bv[kFcBatch];
#pragma unroll
            for (int u = 0; u < kFcBatch; ++u)
                if (k0 + 16 * u < a.H) bv[u] = *reinterpret_cast<const f32x4*>(wrow + k0 + 16 * u);
#pragma unroll
            for (int u = 0; u < kFcBatch; ++u) {
                if (k0 + 16 * u >= a.H) break;
                const f32x4 av = *reinterpret_cast<const f32x4*>(hrow + k0 + 16 * u);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[u][i], acc, 0, 0, 0);
            }
        }
        if (tile * 16 + r < a.N2) {
            const float bias = a.b2 ? a.b2[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 4 * q + i;
                if (m < a.M) a.y[(size_t)m * a.N2 + n] = acc[i] + bias;
            }
        }
    }
    KA_FTL(6);
#undef KA_FTL
    if (touch == 1.2345678e38f && a.hidden_out) a.hidden_out[0] = touch;        // (never true: see `touch`)
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s*n + i]
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int nsplit, size_t n,
                                    int accumulate) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {                      // (eight partials in flight, added in order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; ++k) s += slab[(size_t)k * n + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}

// partial[z][n] = sum over rows of slice z of A[m][n]  (bias gradients / BN sums over rows)
// with optional second output partial2 = sum A[m][n]*Bm[m][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                     float* __restrict__ part, float* __restrict__ part2, int M,
                                                     int N, int rows_per) {
    __shared__ float red[2][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int mbeg = blockIdx.y * rows_per, mend = min(M, mbeg + rows_per);
    float s = 0.f, s2 = 0.f;
    if (n < N) {
        // eight rows are requested before any is added (same order of additions as the plain loop, 8x the loads in flight)
        int m = mbeg + sl;
        for (; m + 28 < mend; m += 32) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = A[(size_t)(m + 4 * u) * N + n];
                b[u] = Bm ? Bm[(size_t)(m + 4 * u) * N + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; s2 += a[u] * b[u]; }
        }
        for (; m < mend; m += 4) {
            const float a = A[(size_t)m * N + n];
            s += a;
            if (Bm) s2 += a * Bm[(size_t)m * N + n];
        }
    }
    red[0][sl][threadIdx.x & 63] = s; red[1][sl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (sl == 0 && n < N) {
        const int l = threadIdx.x;
        part[(size_t)blockIdx.y * N + n] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        if (part2) part2[(size_t)blockIdx.y * N + n] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}

// g[i] = (h[i] > 0) ? g[i] : 0
__global__ void relu_mask_kernel(float* __restrict__ g, const float* __restrict__ h, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!(h[i] > 0.f)) g[i] = 0.f;
}

// rows of fp32 (M,N): out = relu(in*scale[n]+shift[n])           (policy head BN + ReLU)
__global__ void rows_affine_relu_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                        const float* __restrict__ shift, float* __restrict__ out, size_t total, int N) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % N;
        out[i] = fmaxf(in[i] * scale[n] + shift[n], 0.f);
    }
}
// policy-head BN backward pieces on fp32 rows
//   mode 0: da = dr * [in*scale+shift > 0]                       (in place on dr)
//   mode 1: dy = k1*da + k2 + k3*in                              (in place on dr)
__global__ void rows_bn_bwd_kernel(float* __restrict__ dr, const float* __restrict__ in, const float* __restrict__ p0,
                                   const float* __restrict__ p1, size_t total, int N, int mode) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % N;
        if (mode == 0) { if (!(in[i] * p0[n] + p1[n] > 0.f)) dr[i] = 0.f; }
        else dr[i] = p0[n] * dr[i] + p0[N + n] + p0[2 * N + n] * in[i];
    }
}
// yhat-weighted column sums for the policy BN: part[z][n] = sum da, part2[z][n] = sum da*(in-mean)*invstd
__global__ __launch_bounds__(256) void rows_bn_sums_kernel(const float* __restrict__ da, const float* __restrict__ in,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ part, float* __restrict__ part2, int M,
                                                           int N, int rows_per) {
    __shared__ float red[2][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int mbeg = blockIdx.y * rows_per, mend = min(M, mbeg + rows_per);
    float s = 0.f, s2 = 0.f;
    if (n < N) {
        const float mu = mean[n], is = invstd[n];
        int m = mbeg + sl;
        for (; m + 28 < mend; m += 32) {                 // eight rows in flight (same order of additions)
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = da[(size_t)(m + 4 * u) * N + n]; b[u] = in[(size_t)(m + 4 * u) * N + n]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; s2 += a[u] * ((b[u] - mu) * is); }
        }
        for (; m < mend; m += 4) {
            const float a = da[(size_t)m * N + n];
            s += a; s2 += a * ((in[(size_t)m * N + n] - mu) * is);
        }
    }
    red[0][sl][threadIdx.x & 63] = s; red[1][sl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (sl == 0 && n < N) {
        const int l = threadIdx.x;
        part[(size_t)blockIdx.y * N + n] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        part2[(size_t)blockIdx.y * N + n] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}
// column sum and sum of squares of fp32 rows: part[z][n], part2[z][n]
__global__ __launch_bounds__(256) void rows_sq_sums_kernel(const float* __restrict__ A, float* __restrict__ part,
                                                           float* __restrict__ part2, int M, int N, int rows_per) {
    __shared__ float red[2][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int mbeg = blockIdx.y * rows_per, mend = min(M, mbeg + rows_per);
    float s = 0.f, s2 = 0.f;
    if (n < N) {
        int m = mbeg + sl;
        for (; m + 28 < mend; m += 32) {                 // eight rows in flight (same order of additions)
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = A[(size_t)(m + 4 * u) * N + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; s2 += a[u] * a[u]; }
        }
        for (; m < mend; m += 4) { const float a = A[(size_t)m * N + n]; s += a; s2 += a * a; }
    }
    red[0][sl][threadIdx.x & 63] = s; red[1][sl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (sl == 0 && n < N) {
        const int l = threadIdx.x;
        part[(size_t)blockIdx.y * N + n] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        part2[(size_t)blockIdx.y * N + n] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}

inline int grid1d(size_t n, int cap) { size_t b = (n + 255) / 256; return (int)(b < (size_t)cap ? (b ? b : 1) : cap); }

}  // namespace

// nsplit > 1: C must hold nsplit slabs of M*ldc floats (no bias/relu/bf16 then); reduce with ka_reduce_slabs.
extern "C" int ka_gemm(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, int a_bf16, int b_bf16, int c_bf16, int relu, int accumulate,
                       int nsplit, void* stream) {
    KA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nsplit >= 1, "gemm: bad arguments");
    KA_REQUIRE(nsplit == 1 || (!bias && !relu && !c_bf16 && !accumulate), "gemm: split-K output must be raw fp32 slabs");
    KA_REQUIRE(!(accumulate && c_bf16), "gemm: accumulate needs an fp32 output");
    int len = (K + nsplit - 1) / nsplit;
    len = (len + 31) / 32 * 32;
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, a_bf16, b_bf16, c_bf16, relu, len, accumulate};
    dim3 grid((N + 63) / 64, (M + 63) / 64, nsplit);
    KA_REQUIRE(grid.y <= 65535, "gemm: M too large for grid.y (%d rows)", M);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto rows_aligned = [](const void* p, int ld, int bf16) {
        return ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & (bf16 ? 7 : 15)) == 0;
    };
    g.a_vec = rows_aligned(A, lda, a_bf16);
    g.b_vec = rows_aligned(B, ldb, b_bf16);
    // few workgroups: 16 waves per workgroup shorten the per-wave MFMA chain (latency-bound small-batch GEMMs)
    const bool wide = (long long)grid.x * grid.y * grid.z < 128;
#define KA_GEMM_LAUNCH(TA_, TB_) \
    do { if (wide) hipLaunchKernelGGL((gemm_f32_fast_kernel<TA_, TB_, 16>), grid, dim3(1024), 0, st, g); \
         else hipLaunchKernelGGL((gemm_f32_fast_kernel<TA_, TB_, 4>), grid, dim3(256), 0, st, g); } while (0)
    if (transA && transB) KA_GEMM_LAUNCH(true, true);
    else if (transA) KA_GEMM_LAUNCH(true, false);
    else if (transB) KA_GEMM_LAUNCH(false, true);
    else KA_GEMM_LAUNCH(false, false);
#undef KA_GEMM_LAUNCH
    return ka_check_launch("gemm");
}

// table: device int64 [njobs][10] = {dY, X, dW, db or 0, M, N, K, ldx, x_bf16, first workgroup}; total_wgs = sum over jobs of
// ceil(N/64) * ceil((K + (db != 0)) / 64)
extern "C" int ka_gemm_grouped_wgrad(const long long* table, int njobs, int total_wgs, void* stream) {
    KA_REQUIRE(table && njobs > 0 && total_wgs > 0, "gemm_grouped_wgrad: bad arguments");
    hipLaunchKernelGGL(gemm_grouped_wgrad_kernel, dim3(total_wgs), dim3(256), 0, static_cast<hipStream_t>(stream), table, njobs);
    return ka_check_launch("gemm_grouped_wgrad");
}

// 1 when ka_fc_chain handles the shape (otherwise callers issue the two GEMMs)
extern "C" int ka_fc_chain_supported(int K1, int ldx, int H, int N2) {
    const int T1 = H / 16;
    const int ksplit = T1 >= 8 ? 1 : (T1 > 0 ? 8 / T1 : 0);
    return K1 >= 16 && K1 % 16 == 0 && K1 <= 2048 && ldx % 4 == 0 && ldx >= K1 && H >= 16 && H % 16 == 0 && H <= 512 &&
           (T1 >= 8 || T1 == 1 || T1 == 2 || T1 == 4) && (K1 / ksplit) % 16 == 0 && N2 >= 1;
}

extern "C" int ka_fc_chain(const float* x, const float* in_scale, const float* in_shift, float in_alpha, const float* W1,
                           const float* b1, const float* W2, const float* b2, float* x_out, float* hidden_out, float* y,
                           int M, int K1, int ldx, int H, int N2, void* stream) {
    KA_REQUIRE(x && W1 && W2 && y && M > 0, "fc_chain: null tensor");
    KA_REQUIRE(ka_fc_chain_supported(K1, ldx, H, N2), "fc_chain: unsupported shape (K1=%d ldx=%d H=%d N2=%d)", K1, ldx, H, N2);
    KA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "fc_chain: in_scale / in_shift come together");
    KA_REQUIRE(!x_out || in_scale, "fc_chain: x_out is the transformed input; without a transform pass x itself");
    KA_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(W1) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(W2) & 15) == 0, "fc_chain: operands must be 16-byte aligned");
    const int T1 = H / 16, ksplit = T1 >= 8 ? 1 : 8 / T1;
    const size_t lds = (size_t)(16 * (K1 + 4) + 16 * (H + 4) + (ksplit > 1 ? ksplit * 16 * H : 0)) * sizeof(float);
    KA_REQUIRE(lds <= 160 * 1024, "fc_chain: LDS %zu B", lds);
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&fc_chain_kernel), attr_done, "fc_chain")) return rc;
    ChainArgs a{x, in_scale, in_shift, in_alpha, W1, b1, W2, b2, x_out, hidden_out, y, M, K1, ldx, H, N2, nullptr, ka_debug_stamps().load()};
    hipLaunchKernelGGL(fc_chain_kernel, dim3((M + 15) / 16), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("fc_chain");
}

// Backward of a two-layer chain y = W2 relu(W1 x + b1) + b2 with respect to x, in ONE launch:
//     dhidden = (dy W2) * [hidden > 0]   (written to dhidden_out: the dY of W1's weight gradient)
//     dx      = dhidden W1
// W2T (H, N2) and W1T (K1, H) are the TRANSPOSED weights (ka_transpose_multi keeps such copies current): the kernel is the
// forward chain kernel with the ReLU replaced by the saved-activation mask -- with nn.Linear's own [out, in] layout the
// backward contraction runs along the strided axis (a fused form on that layout was slower than two GEMM launches).
// Replaces autograd's input gradients of global_fc (se_resnet.py:62-63, 71): GEMM + mask + GEMM launches.
extern "C" int ka_fc_chain_bwd(const float* dy, const float* hidden, const float* W2T, const float* W1T, float* dhidden_out,
                               float* dx, int M, int N2, int H, int K1, void* stream) {
    KA_REQUIRE(dy && hidden && W2T && W1T && dhidden_out && dx && M > 0, "fc_chain_bwd: null tensor");
    // as a forward chain: input dy (M, N2) -> "hidden" width H -> output width K1
    KA_REQUIRE(ka_fc_chain_supported(N2, N2, H, K1), "fc_chain_bwd: unsupported shape (N2=%d H=%d K1=%d)", N2, H, K1);
    KA_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(W2T) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(W1T) & 15) == 0, "fc_chain_bwd: operands must be 16-byte aligned");
    const int T1 = H / 16, ksplit = T1 >= 8 ? 1 : 8 / T1;
    const size_t lds = (size_t)(16 * (N2 + 4) + 16 * (H + 4) + (ksplit > 1 ? ksplit * 16 * H : 0)) * sizeof(float);
    KA_REQUIRE(lds <= 160 * 1024, "fc_chain_bwd: LDS %zu B", lds);
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&fc_chain_kernel), attr_done, "fc_chain_bwd")) return rc;
    ChainArgs a{dy, nullptr, nullptr, 1.f, W2T, nullptr, W1T, nullptr, nullptr, dhidden_out, dx, M, N2, N2, H, K1, hidden, ka_debug_stamps().load()};
    hipLaunchKernelGGL(fc_chain_kernel, dim3((M + 15) / 16), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("fc_chain_bwd");
}

namespace {
// table[n][4] (int64) = {src, dst, rows, cols}: dst (cols, rows) = src (rows, cols)^T, fp32; blockIdx.y picks the matrix
__global__ __launch_bounds__(256) void transpose_multi_kernel(const long long* __restrict__ table) {
    __shared__ float tile[32][33];
    const long long* t = table + (size_t)blockIdx.y * 4;
    const KA_GLOBAL float* src = (const KA_GLOBAL float*)t[0];
    KA_GLOBAL float* dst = (KA_GLOBAL float*)t[1];
    const int R = (int)t[2], Cc = (int)t[3];
    const int tiles_c = (Cc + 31) / 32, ntiles = ((R + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
            const int rr = r0 + ty + k, cc = c0 + tx;
            tile[ty + k][tx] = (rr < R && cc < Cc) ? src[(size_t)rr * Cc + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
            const int cc = c0 + ty + k, rr = r0 + tx;
            if (cc < Cc && rr < R) dst[(size_t)cc * R + rr] = tile[tx][ty + k];
        }
    }
}
}  // namespace

// n transposes in one launch (the transposed FC weight copies of ka_fc_chain_bwd, refreshed after an optimiser step)
extern "C" int ka_transpose_multi(const void* table, int n, int max_tiles, void* stream) {
    KA_REQUIRE(table && n > 0 && max_tiles > 0, "transpose_multi: bad arguments");
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(max_tiles < 64 ? max_tiles : 64, n), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const long long*>(table));
    return ka_check_launch("transpose_multi");
}

extern "C" int ka_reduce_slabs(const float* slab, float* out, int nsplit, long long n, int accumulate, void* stream) {
    KA_REQUIRE(slab && out && nsplit >= 1 && n > 0, "reduce_slabs: bad arguments");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid1d((size_t)n, 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), slab, out, nsplit, (size_t)n, accumulate);
    return ka_check_launch("reduce_slabs");
}

// two slab sets of one split count in one launch (a linear layer's weight- and bias-gradient partials): out_a[i] = sum_s slab_a[s*na + i],
// out_b likewise -- the same sums in the same order as two ka_reduce_slabs calls
__global__ void reduce_slabs2_kernel(const float* __restrict__ sa, float* __restrict__ oa, size_t na, const float* __restrict__ sb,
                                     float* __restrict__ ob, size_t nb, int nsplit) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
        const bool first = i < na;
        const float* __restrict__ sl = first ? sa : sb;
        const size_t n = first ? na : nb, j = first ? i : i - na;
        // (eight partials requested before the first is added: as a rolled loop every split was an L2 round trip of its own;
        //  the additions keep their order)
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sl[(size_t)(k + u) * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; ++k) s += sl[(size_t)k * n + j];
        (first ? oa : ob)[j] = s;
    }
}
extern "C" int ka_reduce_slabs2(const float* slab_a, float* out_a, long long na, const float* slab_b, float* out_b, long long nb,
                                int nsplit, void* stream) {
    KA_REQUIRE(slab_a && out_a && slab_b && out_b && nsplit >= 1 && na > 0 && nb > 0, "reduce_slabs2: bad arguments");
    hipLaunchKernelGGL(reduce_slabs2_kernel, dim3(grid1d((size_t)(na + nb), 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slab_a, out_a, (size_t)na, slab_b, out_b, (size_t)nb, nsplit);
    return ka_check_launch("reduce_slabs2");
}

// part (and part2) must hold nsplit*N floats; rows are split evenly over nsplit slices
extern "C" int ka_colsum(const float* A, const float* Bm, float* part, float* part2, int M, int N, int nsplit,
                         void* stream) {
    KA_REQUIRE(A && part && M > 0 && N > 0 && nsplit >= 1, "colsum: bad arguments");
    const int rows_per = (M + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, nsplit), dim3(256), 0, static_cast<hipStream_t>(stream), A, Bm,
                       part, part2, M, N, rows_per);
    return ka_check_launch("colsum");
}

extern "C" int ka_relu_mask(float* g, const float* h, long long n, void* stream) {
    KA_REQUIRE(g && h && n > 0, "relu_mask: bad arguments");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid1d((size_t)n, 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g, h, (size_t)n);
    return ka_check_launch("relu_mask");
}

extern "C" int ka_rows_affine_relu(const float* in, const float* scale, const float* shift, float* out, long long M,
                                   int N, void* stream) {
    KA_REQUIRE(in && scale && shift && out, "rows_affine_relu: null tensor");
    hipLaunchKernelGGL(rows_affine_relu_kernel, dim3(grid1d((size_t)M * N, 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, scale, shift, out, (size_t)M * N, N);
    return ka_check_launch("rows_affine_relu");
}

extern "C" int ka_rows_bn_bwd(float* dr, const float* in, const float* p0, const float* p1, long long M, int N,
                              int mode, void* stream) {
    KA_REQUIRE(dr && in && p0 && (mode == 1 || p1), "rows_bn_bwd: null tensor");
    hipLaunchKernelGGL(rows_bn_bwd_kernel, dim3(grid1d((size_t)M * N, 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dr, in, p0, p1, (size_t)M * N, N, mode);
    return ka_check_launch("rows_bn_bwd");
}

extern "C" int ka_rows_bn_sums(const float* da, const float* in, const float* mean, const float* invstd, float* part,
                               float* part2, int M, int N, int nsplit, void* stream) {
    KA_REQUIRE(da && in && mean && invstd && part && part2, "rows_bn_sums: null tensor");
    const int rows_per = (M + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(rows_bn_sums_kernel, dim3((N + 63) / 64, nsplit), dim3(256), 0, static_cast<hipStream_t>(stream),
                       da, in, mean, invstd, part, part2, M, N, rows_per);
    return ka_check_launch("rows_bn_sums");
}

extern "C" int ka_rows_sq_sums(const float* A, float* part, float* part2, int M, int N, int nsplit, void* stream) {
    KA_REQUIRE(A && part && part2, "rows_sq_sums: null tensor");
    const int rows_per = (M + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(rows_sq_sums_kernel, dim3((N + 63) / 64, nsplit), dim3(256), 0, static_cast<hipStream_t>(stream),
                       A, part, part2, M, N, rows_per);
    return ka_check_launch("rows_sq_sums");
}
