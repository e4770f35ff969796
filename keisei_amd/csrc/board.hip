// Per-board kernels around the convolutions: BatchNorm statistics/coefficients, the fused
// SE + residual + ReLU + global-pool tail of GlobalPoolBiasBlock, and their backward passes.
//
// All are HBM-bound streaming kernels over NHWC (B,81,C) activations.  One workgroup owns one
// board; a thread owns 2 adjacent channels (4-byte bf16x2 / 8-byte f32x2 accesses, a wave
// covers a contiguous 256/512-byte run along the channel axis) and a strided subset of the
// 81 squares, which it keeps in registers, so every per-(board,channel) reduction -- global
// pool mean/max/std, SE squeeze, the BN partial sums -- is a register loop plus one small LDS
// combine, and two-pass statistics (std, tie counts) never re-read HBM.
//
// Reference semantics: keisei/training/models/se_resnet.py:68-98 (forward) and the autograd
// derivatives of mean / amax (ties share the gradient equally) / std(correction=0) (zero where
// sigma == 0) / BatchNorm2d (training mode) / sigmoid gate.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPPT = 41;     // squares per thread when 2 square-slices are used (81/2)

struct BoardMap {
    int cpw;      // channel pairs handled per pass
    int ph;       // square slices
    int cp;       // this thread's channel pair within the pass
    int slice;    // this thread's square slice
    bool active;
    __device__ BoardMap(int C) {
        const int pairs = C >> 1;
        cpw = pairs < 128 ? pairs : 128;
        ph = kThreads / cpw;
        if (ph > KA_BOARD) ph = KA_BOARD;
        cp = threadIdx.x % cpw;
        slice = threadIdx.x / cpw;
        active = slice < ph;
    }
};

// combine per-slice partials: red[slice][cpw*2]; returns total for this thread's pair (all threads)
__device__ __forceinline__ f32x2 combine_sum(float* red, const BoardMap& m, f32x2 v) {
    __syncthreads();
    if (m.active) { red[(m.slice * m.cpw + m.cp) * 2] = v[0]; red[(m.slice * m.cpw + m.cp) * 2 + 1] = v[1]; }
    __syncthreads();
    f32x2 t = {0.f, 0.f};
    for (int s = 0; s < m.ph; ++s) { t[0] += red[(s * m.cpw + m.cp) * 2]; t[1] += red[(s * m.cpw + m.cp) * 2 + 1]; }
    return t;
}
__device__ __forceinline__ f32x2 combine_max(float* red, const BoardMap& m, f32x2 v) {
    __syncthreads();
    if (m.active) { red[(m.slice * m.cpw + m.cp) * 2] = v[0]; red[(m.slice * m.cpw + m.cp) * 2 + 1] = v[1]; }
    __syncthreads();
    f32x2 t = {-INFINITY, -INFINITY};
    for (int s = 0; s < m.ph; ++s) {
        t[0] = fmaxf(t[0], red[(s * m.cpw + m.cp) * 2]);
        t[1] = fmaxf(t[1], red[(s * m.cpw + m.cp) * 2 + 1]);
    }
    return t;
}

// ---------------------------------------------------------------- input layout
// obs (B,Cobs,9,9) f32 NCHW -> (B,81,Cpad) T NHWC, channels >= Cobs zero
template <typename T>
__global__ void obs_to_nhwc_kernel(const float* __restrict__ obs, const long long* __restrict__ idx,
                                   T* __restrict__ out, int B, int Cobs, int Cpad) {
    __shared__ float tile[KA_BOARD * 65];
    const int b = blockIdx.x;
    const long long sb = idx ? idx[b] : b;      // fused minibatch gather (katago_ppo.py:835)
    for (int c0 = 0; c0 < Cpad; c0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * KA_BOARD; i += blockDim.x) {
            const int c = i / KA_BOARD, p = i - c * KA_BOARD;
            tile[p * 65 + c] = (c0 + c < Cobs) ? obs[((size_t)sb * Cobs + c0 + c) * KA_BOARD + p] : 0.f;
        }
        __syncthreads();
        const int w = min(64, Cpad - c0);
        for (int i = threadIdx.x; i < KA_BOARD * w; i += blockDim.x) {
            const int p = i / w, c = i - p * w;
            Elem<T>::st(out + ((size_t)b * KA_BOARD + p) * Cpad + c0 + c, tile[p * 65 + c]);
        }
    }
}

// (B,81,C) T NHWC -> (B,C,9,9) f32 NCHW (API boundary of a stand-alone block)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int C) {
    __shared__ float tile[KA_BOARD * 65];
    const int b = blockIdx.x;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int w = min(64, C - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < KA_BOARD * w; i += blockDim.x) {
            const int p = i / w, c = i - p * w;
            tile[p * 65 + c] = Elem<T>::ld(in + ((size_t)b * KA_BOARD + p) * C + c0 + c);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < w * KA_BOARD; i += blockDim.x) {
            const int c = i / KA_BOARD, p = i - c * KA_BOARD;
            out[((size_t)b * C + c0 + c) * KA_BOARD + p] = tile[p * 65 + c];
        }
    }
}

// ---------------------------------------------------------------- BatchNorm statistics
// Two-stage, fixed-order fp64 column sums of two fp32 planes (A: rowsA x C, B: rowsB x C):
//   stage 1: grid (C/64, kRedSlices) -> part[slice][2C];  stage 2: sums[0:C] = sum_s part[s][c], sums[C:2C] likewise.
// Used for BN forward statistics (per-board sums + per-workgroup squares from the conv epilogue) and for the BN
// backward sums (two per-board planes).
constexpr int kRedSlices = 64;
__device__ __forceinline__ void colsum2_stage1_body(const float* __restrict__ A, int rowsA, const float* __restrict__ Bp, int rowsB,
                                                    int C, double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int slice = blockIdx.y * 4 + sub, nsl = kRedSlices * 4;
    __shared__ double red[2][4][64];
    double s = 0.0, ss = 0.0;
    if (c < C) {
        // eight rows of each operand requested before the first is added (same order of additions: the loop was a chain of
        // L2 round trips)
        int r = slice;
        for (; r + 15 * nsl < rowsA && r + 15 * nsl < rowsB; r += 16 * nsl) {      // (4096 rows over 256 slices: all sixteen at once)
            float va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { va[u] = A[(size_t)(r + u * nsl) * C + c]; vb[u] = Bp[(size_t)(r + u * nsl) * C + c]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { s += (double)va[u]; ss += (double)vb[u]; }
        }
        for (; r + 7 * nsl < rowsA && r + 7 * nsl < rowsB; r += 8 * nsl) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { va[u] = A[(size_t)(r + u * nsl) * C + c]; vb[u] = Bp[(size_t)(r + u * nsl) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)va[u]; ss += (double)vb[u]; }
        }
        for (int ra = r; ra < rowsA; ra += nsl) s += (double)A[(size_t)ra * C + c];
        for (int rb = r; rb < rowsB; rb += nsl) ss += (double)Bp[(size_t)rb * C + c];
    }
    red[0][sub][threadIdx.x & 63] = s;
    red[1][sub][threadIdx.x & 63] = ss;
    __syncthreads();
    if (sub == 0 && c < C) {
        const int l = threadIdx.x;
        part[(size_t)blockIdx.y * 2 * C + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        part[(size_t)blockIdx.y * 2 * C + C + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}
__global__ __launch_bounds__(256) void colsum2_stage1_kernel(const float* __restrict__ A, int rowsA,
                                                             const float* __restrict__ Bp, int rowsB, int C,
                                                             double* __restrict__ part) {
    colsum2_stage1_body(A, rowsA, Bp, rowsB, C, part);
}
// Stage 1 and the coefficient kernel that follows it in ONE launch: the workgroup that finishes a 64-channel column group LAST
// (an arrival counter per group, reset by that workgroup for the next launch) computes the group's coefficients from the 64
// partial rows, in slice order -- so the result does not depend on which workgroup that is.  The partials cross XCDs (each has
// its own L2): every workgroup publishes its row with a device-scope release fence before it arrives, the last one acquires
// before it reads.  164 launches (and as many launch boundaries) less per training step.
__device__ __forceinline__ bool colsum2_arrive_last(int* __restrict__ counters) {
    __shared__ int is_last;
    __syncthreads();                                         // this workgroup's partial row is written
    if (threadIdx.x == 0) {
        __threadfence();                                     // release: the row is visible device-wide before the arrival is
        const int prev = atomicAdd(&counters[blockIdx.x], 1);
        is_last = prev == (int)gridDim.y - 1;
        if (is_last) counters[blockIdx.x] = 0;               // (nobody else touches it until the next launch)
    }
    __syncthreads();
    if (!is_last) return false;
    __threadfence();                                         // acquire: the other workgroups' rows
    return true;
}

// training-mode coefficients: y_hat*gamma+beta == x*scale+shift.  Updates running stats like
// nn.BatchNorm2d (momentum form, unbiased running variance).
// `sums` is either the reduced [2C] vector (nparts == 1) or the stage-1 partials [nparts][2C], summed here in slice order
// (a thread owns a channel: the 64 partials are fetched 16 at a time BEFORE they are added, in slice order -- a plain
// `s += p[k]` loop issues one L2 round trip per element and made these 4-workgroup kernels take 35 us each)
__device__ __forceinline__ double sum_parts(const double* __restrict__ p, int nparts, int stride, int c) {
    double s = 0.0;
    int k = 0;
    // (all kRedSlices partials of a sum requested before the first is added: in batches of 16 a coefficient kernel was eight dependent
    //  L2 round trips long -- 6.4 us for a 256-channel layer, 328 such launches per step)
    for (; k + 64 <= nparts; k += 64) {
        double v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = p[(size_t)(k + u) * stride + c];
#pragma unroll
        for (int u = 0; u < 64; ++u) s += v[u];
    }
    for (; k + 16 <= nparts; k += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(k + u) * stride + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < nparts; ++k) s += p[(size_t)k * stride + c];
    return s;
}
__global__ void colsum2_stage2_kernel(const double* __restrict__ part, double* __restrict__ sums, int C2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C2) return;
    sums[c] = sum_parts(part, kRedSlices, C2, c);      // (16 partials in flight at a time, slice order)
}

// SyncBatchNorm form of stage 2: the vector that travels through the cross-rank all-reduce is [2C sums | element count],
// so the count is written here (no separate fill launch, no host-side scalar store), and the backward keeps an
// un-reduced copy of the local sums (dgamma / dbeta are per-rank) without a clone launch.
__global__ void colsum2_stage2_sync_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                           double* __restrict__ local_copy, double count, int C2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > C2) return;
    double s = count;
    if (c < C2) {
        s = sum_parts(part, kRedSlices, C2, c);        // (16 partials in flight at a time, slice order)
    }
    sums[c] = s;
    if (local_copy) local_copy[c] = s;
}

__device__ __forceinline__ void bn_coeffs_body(int c, const double* __restrict__ sums, int nparts, double count, const double* __restrict__ count_dev,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                               float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                               float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                               float* __restrict__ invstd_out, int C) {
    // (inlined into two kernels: without this the compiler is free to contract a*b+c differently in each, and the running
    //  statistics of the one-launch and two-launch paths drift apart by an ulp)
#pragma clang fp contract(off)
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    if (count_dev) count = *count_dev;          // global element count after a cross-rank all-reduce
    const double mean = sum_parts(sums, nparts, 2 * C, c) / count;
    double var = sum_parts(sums, nparts, 2 * C, C + c) / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void bn_coeffs_kernel(const double* __restrict__ sums, int nparts, double count, const double* __restrict__ count_dev,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                 float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                 float* __restrict__ invstd_out, int C) {
    bn_coeffs_body(blockIdx.x * blockDim.x + threadIdx.x, sums, nparts, count, count_dev, gamma, beta, running_mean, running_var,
                   num_batches_tracked, momentum, eps, scale, shift, mean_out, invstd_out, C);
}
__global__ __launch_bounds__(256) void colsum2_bn_coeffs_kernel(const float* __restrict__ A, int rowsA, const float* __restrict__ Bp, int rowsB,
                                                                int C, double* part, int* __restrict__ counters, double count,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                                                float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    colsum2_stage1_body(A, rowsA, Bp, rowsB, C, part);
    if (!colsum2_arrive_last(counters)) return;
    if (threadIdx.x < 64)
        bn_coeffs_body(blockIdx.x * 64 + threadIdx.x, part, kRedSlices, count, nullptr, gamma, beta, running_mean, running_var,
                       num_batches_tracked, momentum, eps, scale, shift, mean_out, invstd_out, C);
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// every BatchNorm layer of the model in one launch (rollout inference): table rows {gamma, beta, running_mean,
// running_var, scale, shift, C, eps (float bits)}; blockIdx.y = layer
__global__ void bn_eval_coeffs_multi_kernel(const long long* __restrict__ table) {
    const long long* t = table + (size_t)blockIdx.y * 8;
    const int C = (int)t[6];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* gamma = reinterpret_cast<const float*>(t[0]);
    const float* beta = reinterpret_cast<const float*>(t[1]);
    const float* rm = reinterpret_cast<const float*>(t[2]);
    const float* rv = reinterpret_cast<const float*>(t[3]);
    const float eps = __uint_as_float((unsigned int)t[7]);
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    reinterpret_cast<float*>(t[4])[c] = sc;
    reinterpret_cast<float*>(t[5])[c] = beta[c] - rm[c] * sc;
}

// dgamma/dbeta from the LOCAL sums; dy = k1*dz + k2 + k3*y from the (possibly all-reduced) sums
__device__ __forceinline__ void bn_bwd_coeffs_body(int c, const double* __restrict__ sums_local, const double* __restrict__ sums_global,
                                                   int nparts, double count, const double* __restrict__ count_dev,
                                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, float* __restrict__ k, int C, int train) {
#pragma clang fp contract(off)
    if (c >= C) return;
    if (count_dev) count = *count_dev;
    // nparts > 1: both pointers are the same stage-1 partials (no cross-rank reduction in between)
    const double l1 = sum_parts(sums_local, nparts, 2 * C, c), l2 = sum_parts(sums_local, nparts, 2 * C, C + c);
    const double g1 = nparts > 1 ? l1 : sums_global[c], g2 = nparts > 1 ? l2 : sums_global[C + c];
    if (dgamma) { dbeta[c] = (float)l1; dgamma[c] = (float)l2; }
    const double g = gamma[c], is = invstd[c], mu = mean[c];
    const double k1 = g * is;
    // eval mode (running statistics are constants): dy = gamma*invstd*dz
    const double k3 = train ? -g * is * is * g2 / count : 0.0;
    const double k2 = train ? -g * is * g1 / count - k3 * mu : 0.0;
    k[c] = (float)k1; k[C + c] = (float)k2; k[2 * C + c] = (float)k3;
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void bn_bwd_coeffs_kernel(const double* __restrict__ sums_local, const double* __restrict__ sums_global,
                                     int nparts, double count, const double* __restrict__ count_dev,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ k, int C, int train) {
    bn_bwd_coeffs_body(blockIdx.x * blockDim.x + threadIdx.x, sums_local, sums_global, nparts, count, count_dev, gamma, mean, invstd,
                       dgamma, dbeta, k, C, train);
}
__global__ __launch_bounds__(256) void colsum2_bn_bwd_coeffs_kernel(const float* __restrict__ A, const float* __restrict__ Bp, int rows, int C,
                                                                    double* part, int* __restrict__ counters, double count,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ k, int train) {
    colsum2_stage1_body(A, rows, Bp, rows, C, part);
    if (!colsum2_arrive_last(counters)) return;
    if (threadIdx.x < 64)
        bn_bwd_coeffs_body(blockIdx.x * 64 + threadIdx.x, part, part, kRedSlices, count, nullptr, gamma, mean, invstd, dgamma, dbeta, k, C, train);
}

// out[b,c] = a[c]*in[b,c]*mul + s[c]     (SE squeeze of the normalised conv2 output)
__global__ void affine_rows_kernel(const float* __restrict__ in, const float* __restrict__ a,
                                   const float* __restrict__ s, float mul, float* __restrict__ out, int B, int C) {
    const size_t n = (size_t)B * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        out[i] = a[c] * (in[i] * mul) + s[c];
    }
}

// dy = k1[c]*dz + k2[c] + k3[c]*y   (BatchNorm backward, elementwise part).  One workgroup per board, a thread keeps
// its two channels' coefficients in registers and walks its squares (no per-element index arithmetic).
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                                const float* __restrict__ k, T* __restrict__ dy, int C) {
    const BoardMap m(C);
    if (!m.active) return;
    const size_t base = (size_t)blockIdx.x * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        const f32x2 k1 = {k[c], k[c + 1]}, k2 = {k[C + c], k[C + c + 1]}, k3 = {k[2 * C + c], k[2 * C + c + 1]};
        for (int p = m.slice; p < KA_BOARD; p += m.ph) {
            const size_t off = base + (size_t)p * C + c;
            const f32x2 g = ld2(dz + off), v = ld2(y + off);
            st2(dy + off, f32x2{k1[0] * g[0] + k2[0] + k3[0] * v[0], k1[1] * g[1] + k2[1] + k3[1] * v[1]});
        }
    }
}

// ---------------------------------------------------------------- forward tail
// One-pass running statistics of a thread's squares for 2 adjacent channels: Welford mean/M2 (exactly 0 for
// a constant plane), running max and the number of squares that attain it (autograd's amax backward shares the
// gradient between ties).  Slices are merged in a fixed order with Chan's formula.
struct PoolStat {
    f32x2 mean = {0.f, 0.f}, m2 = {0.f, 0.f}, mx = {-INFINITY, -INFINITY}, cnt = {0.f, 0.f};
    float n = 0.f;
    __device__ __forceinline__ void push(const f32x2 x) {
        n += 1.f;
        const float inv = 1.f / n;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float d = x[e] - mean[e];
            mean[e] += d * inv;
            m2[e] += d * (x[e] - mean[e]);
            if (x[e] > mx[e]) { mx[e] = x[e]; cnt[e] = 1.f; }
            else if (x[e] == mx[e]) cnt[e] += 1.f;
        }
    }
};
// red: [slice][cpw][9]; the slice-0 thread of every channel pair merges and writes pool[b] = [mean|max|std|ties] (4C)
__device__ __forceinline__ void pool_merge_store(float* red, const BoardMap& m, const PoolStat& st, float* pool_row,
                                                 int c, int C) {
    __syncthreads();
    if (m.active) {
        float* r = red + (m.slice * m.cpw + m.cp) * 9;
        r[0] = st.n; r[1] = st.mean[0]; r[2] = st.mean[1]; r[3] = st.m2[0]; r[4] = st.m2[1];
        r[5] = st.mx[0]; r[6] = st.mx[1]; r[7] = st.cnt[0]; r[8] = st.cnt[1];
    }
    __syncthreads();
    if (m.active && m.slice == 0 && pool_row) {
        float n = 0.f, mean[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f}, mx[2] = {-INFINITY, -INFINITY}, cnt[2] = {0.f, 0.f};
        for (int s = 0; s < m.ph; ++s) {
            const float* r = red + (s * m.cpw + m.cp) * 9;
            const float nb = r[0];
            if (nb == 0.f) continue;
            const float nn = n + nb;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float d = r[1 + e] - mean[e];
                mean[e] += d * (nb / nn);
                m2[e] += r[3 + e] + d * d * (n * nb / nn);
                if (r[5 + e] > mx[e]) { mx[e] = r[5 + e]; cnt[e] = r[7 + e]; }
                else if (r[5 + e] == mx[e]) cnt[e] += r[7 + e];
            }
            n = nn;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            pool_row[c + e] = mean[e];
            pool_row[C + c + e] = mx[e];
            pool_row[2 * C + c + e] = sqrtf(fmaxf(m2[e], 0.f) / KA_BOARD);
            pool_row[3 * C + c + e] = cnt[e];
        }
    }
}

// out = relu( (scale*y+shift) [* sigmoid(se[b,c]) + se[b,C+c]] [+ res] ),  pool[b] = [mean|max|std|ties] of out (4C)
template <typename T>
__global__ __launch_bounds__(kThreads) void block_tail_fwd_kernel(
    const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ se, const T* __restrict__ res, T* __restrict__ out, float* __restrict__ pool, int C) {
    extern __shared__ float red[];
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        PoolStat st;
        if (m.active) {
            const f32x2 sc = {scale[c], scale[c + 1]}, sh = {shift[c], shift[c + 1]};
            f32x2 gate = {1.f, 1.f}, bias = {0.f, 0.f};
            if (se) {
                gate = f32x2{sigmoidf_(se[(size_t)b * 2 * C + c]), sigmoidf_(se[(size_t)b * 2 * C + c + 1])};
                bias = f32x2{se[(size_t)b * 2 * C + C + c], se[(size_t)b * 2 * C + C + c + 1]};
            }
            for (int p = m.slice; p < KA_BOARD; p += m.ph) {
                const size_t off = base + (size_t)p * C + c;
                f32x2 u = ld2(y + off);
                u[0] = (u[0] * sc[0] + sh[0]) * gate[0] + bias[0];
                u[1] = (u[1] * sc[1] + sh[1]) * gate[1] + bias[1];
                if (res) { const f32x2 rr = ld2(res + off); u[0] += rr[0]; u[1] += rr[1]; }
                u[0] = rnd<T>(fmaxf(u[0], 0.f));
                u[1] = rnd<T>(fmaxf(u[1], 0.f));
                st2(out + off, u);
                st.push(u);
            }
        }
        pool_merge_store(red, m, st, pool ? pool + (size_t)b * 4 * C : nullptr, c, C);
    }
}

// pool[b] = [mean|max|std|ties] of an arbitrary (B,81,C) tensor (se_resnet.py:93-98)
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_fwd_kernel(const T* __restrict__ x, float* __restrict__ pool, int C) {
    extern __shared__ float red[];
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        PoolStat st;
        if (m.active)
            for (int p = m.slice; p < KA_BOARD; p += m.ph) st.push(ld2(x + base + (size_t)p * C + c));
        pool_merge_store(red, m, st, pool + (size_t)b * 4 * C, c, C);
    }
}

// ---------------------------------------------------------------- backward tail, pass 1
// du = dout*[out>0]; z = scale*y+shift; dse[b,c] = sig'(a)*sum_p du*z ; dse[b,C+c] = sum_p du
template <typename T>
__global__ __launch_bounds__(kThreads) void tail_bwd_reduce_kernel(
    const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ se, float* __restrict__ dse, int C) {
    extern __shared__ float red[];
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        f32x2 r1 = {0.f, 0.f}, r2 = {0.f, 0.f};
        if (m.active) {
            const f32x2 sc = {scale[c], scale[c + 1]}, sh = {shift[c], shift[c + 1]};
            for (int p = m.slice; p < KA_BOARD; p += m.ph) {
                const size_t off = base + (size_t)p * C + c;
                const f32x2 g = ld2(dout + off), o = ld2(out + off), yy = ld2(y + off);
                const float d0 = o[0] > 0.f ? g[0] : 0.f, d1 = o[1] > 0.f ? g[1] : 0.f;
                r1[0] += d0 * (yy[0] * sc[0] + sh[0]); r1[1] += d1 * (yy[1] * sc[1] + sh[1]);
                r2[0] += d0; r2[1] += d1;
            }
        }
        const f32x2 t1 = combine_sum(red, m, r1);
        const f32x2 t2 = combine_sum(red, m, r2);
        if (m.active && m.slice == 0) {
            const float s0 = sigmoidf_(se[(size_t)b * 2 * C + c]), s1 = sigmoidf_(se[(size_t)b * 2 * C + c + 1]);
            float* d = dse + (size_t)b * 2 * C;
            d[c] = t1[0] * s0 * (1.f - s0); d[c + 1] = t1[1] * s1 * (1.f - s1);
            d[C + c] = t2[0]; d[C + c + 1] = t2[1];
        }
    }
}

// ---------------------------------------------------------------- backward tail, pass 2
// dz = du*sigmoid(a) + dsq[b,c]/81 ; per-board partials s1 = sum dz, s2 = sum dz*yhat
template <typename T>
__global__ __launch_bounds__(kThreads) void tail_bwd_dz_kernel(
    const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ y, const float* __restrict__ se,
    const float* __restrict__ dsq, const float* __restrict__ mean, const float* __restrict__ invstd,
    T* __restrict__ dz, float* __restrict__ s1p, float* __restrict__ s2p, int C) {
    extern __shared__ float red[];
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
        if (m.active) {
            const f32x2 gate = {sigmoidf_(se[(size_t)b * 2 * C + c]), sigmoidf_(se[(size_t)b * 2 * C + c + 1])};
            const f32x2 add = {dsq[(size_t)b * C + c] / KA_BOARD, dsq[(size_t)b * C + c + 1] / KA_BOARD};
            const f32x2 mu = {mean[c], mean[c + 1]}, is = {invstd[c], invstd[c + 1]};
            for (int p = m.slice; p < KA_BOARD; p += m.ph) {
                const size_t off = base + (size_t)p * C + c;
                const f32x2 g = ld2(dout + off), o = ld2(out + off), yy = ld2(y + off);
                f32x2 d = {(o[0] > 0.f ? g[0] : 0.f) * gate[0] + add[0], (o[1] > 0.f ? g[1] : 0.f) * gate[1] + add[1]};
                st2(dz + off, d);
                a1[0] += d[0]; a1[1] += d[1];
                a2[0] += d[0] * ((yy[0] - mu[0]) * is[0]); a2[1] += d[1] * ((yy[1] - mu[1]) * is[1]);
            }
        }
        const f32x2 t1 = combine_sum(red, m, a1);
        const f32x2 t2 = combine_sum(red, m, a2);
        if (m.active && m.slice == 0) {
            s1p[(size_t)b * C + c] = t1[0]; s1p[(size_t)b * C + c + 1] = t1[1];
            s2p[(size_t)b * C + c] = t2[0]; s2p[(size_t)b * C + c + 1] = t2[1];
        }
    }
}

// ---------------------------------------------------------------- backward tail, single pass
// tail_bwd_reduce + the per-board squeeze-excite FC chain backward + tail_bwd_dz in ONE read of dout/out/y:
//   du = dout*[out>0]                                          (kept in registers between the two phases)
//   dse[b,c] = sig'(a_c) * sum_p du*z, dse[b,C+c] = sum_p du   (z = scale*y+shift)                -> written out
//   dh[b,j]  = [se1[b,j] > 0] * sum_k dse[b,k] * W2[k,j]       (se_fc2 backward + ReLU mask)      -> written out
//   dsq[b,c] = sum_j dh[b,j] * W1[j,c]                         (se_fc1 backward)
//   dz = du*sigmoid(a_c) + dsq[b,c]/81;   s1 = sum_p dz, s2 = sum_p dz*yhat
// One workgroup per board; a thread owns one 16-byte channel piece (8 bf16 / 4 f32 channels) and every nsl-th square.
// Everything the second phase needs from y is linear in three per-(board, channel) sums taken in the first:
//   Sg = sum_p du, Sgy = sum_p du*(y-mean), Sy = sum_p (y-mean)     (centred on the BatchNorm mean: no cancellation)
//   sum_p du*z = scale*Sgy + (shift + mean*scale)*Sg
//   s1 = gate*Sg + 81*add,  s2 = invstd*(gate*Sgy + add*Sy)         (add = dsq/81)
// so y is not kept: a board costs a thread MAXSQ 16-byte registers instead of 2*MAXSQ, which is what lets three
// workgroups share a CU (the kernel waits on memory: more boards in flight per CU is what it needs), and the second
// reduction through LDS is gone.
// The FC weight gradients (dW2 = dse^T se1, dW1 = dh^T sqz) stay with the GEMM kernels, off the data-gradient chain.
// DX = the block-input gradient of the block ABOVE (block_dx16_kernel) computed in the same pass: this block's `dout` is
// that kernel's result dx and this block's `out` is that block's input x, so the fused launch reads x, dxc, dout', out', y
// (5 tensors) and writes dx, dz instead of reading 4 + 3 and writing 1 + 1: two activation tensors less HBM traffic per
// block.  dx is rounded to T before it is used, so dz / dse / dh / s1 / s2 are bit for bit those of the two launches.
struct DxArgs {
    const void* dxc; const void* dout_up; const void* out_up;   // conv1 data gradient, gradient and output of the block above (null for the heads)
    const float* xpool; const float* dpool;                     // pooled statistics of x, gradient wrt them
    void* dx;
    int du_io;   // 1 (kernel form DX = 2): dout_up already carries its ReLU mask (out_up unused) and dx is WRITTEN masked by [x > 0] -- see ka_block_dx_tail_bwd_du
                 // 2 (kernel form DX = 3): the same, and dz is NOT written: gate_out / add_out carry what forms it (ka_block_dx_tail_bwd_du_gate)
    float* gate_out; float* add_out;                            // [B, C] each: dz = du * gate_out + add_out
};

// DX: 0 = the tail alone, 1 = with the block-input gradient of the block above, 2 = the same in its du-chain form,
// 3 = the du chain WITHOUT dz: dz = du * gate[b, c] + add[b, c] is affine in the du this launch writes anyway, and its only
// reader -- the conv2 data gradient's input transform -- can form it from du and the two per-(board, channel) vectors
// (ka_conv3x3_dgrad_fused_gated).  One activation write less (4 reads + 1 write), du is not kept in registers for a second
// pass, and nothing of a board streams after its FC chain.  s1 / s2 were never sums over the rounded dz (see above), so they
// are unchanged.  (Form 3 at six waves per SIMD -- three boards per CU -- spills at 80 registers: 285 us against 227, and 496
// against 175 once the loads are batched; not kept.)
// Half-width pieces (KA_TAIL_GATE_P4=1, gate form, bf16): a thread owns FOUR channels (8-byte pieces) and every eighth square, so
// its per-channel state -- three sums, the coefficient registers, the unpacked values -- is half as wide: 80 registers, three
// boards per CU, four squares per round trip.
struct ElemBf4 {
    static constexpr int kPer16 = 4;
    typedef bf16x4 vec16;
    static __device__ __forceinline__ void unpack(const vec16& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ vec16 pack(const float* f) {
        vec16 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (__bf16)f[i];
        return v;
    }
};
template <typename T, int MAXSQ, int NTHR, int DX, typename EE = Elem<T>>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(((MAXSQ <= 6 && !DX) || EE::kPer16 * (int)sizeof(T) < 16) ? 6 : 4))) void tail_bwd_fused_kernel(
    const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ se, const float* __restrict__ se1,
    const float* __restrict__ W2, const float* __restrict__ W1, const float* __restrict__ mean,
    const float* __restrict__ invstd, T* __restrict__ dz, float* __restrict__ dse_out, float* __restrict__ dh_out,
    float* __restrict__ s1p, float* __restrict__ s2p, int C, int H, DxArgs dxa) {
    typedef EE E;
    typedef typename E::vec16 vec16;
    constexpr int P16 = E::kPer16;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int groups = C / P16, nsl = NTHR / groups;          // channel pieces per square, square slices
    const int cg = tid % groups, slice = tid / groups, c0 = cg * P16;
    // partial sums: the slices that share a wave are combined by lane exchanges, one LDS row per wave (per slice when a
    // slice is wider than a wave)
    const int rowthr = groups > 64 ? groups : 64, nrow = NTHR / rowthr;
    float* red_g = lds;                      // [nrow][C]
    float* red_gy = red_g + nrow * C;        // [nrow][C]
    float* red_y = red_gy + nrow * C;        // [nrow][C]
    float* v_dse = red_y + nrow * C;         // [2C]
    float* v_part = v_dse + 2 * C;           // [NTHR]
    float* v_dh = v_part + NTHR;             // [H]
    float* v_dsq = v_dh + H;                 // [C]
    float* v_gate = v_dsq + C;               // [C]: sigmoid of the gate logits, one evaluation per channel
    float* v_mu = v_gate + C;                // [C]: BatchNorm mean (read per square: eight registers fewer across the loop)
    float* v_dx = v_mu + C;                  // DX: [5][C] pooling-gradient coefficients of block_dx16_kernel (read per square)
    const size_t base = (size_t)b * KA_BOARD * C + c0;

    for (int c = tid; c < C; c += NTHR) v_mu[c] = mean[c];
    if constexpr (DX) {
        const float* xp = dxa.xpool + (size_t)b * 4 * C;
        const float* dp = dxa.dpool + (size_t)b * 3 * C;
        for (int c = tid; c < C; c += NTHR) {
            const float sd = xp[2 * C + c];
            v_dx[c] = xp[c];
            v_dx[C + c] = xp[C + c];
            v_dx[2 * C + c] = dp[c] / KA_BOARD;
            v_dx[3 * C + c] = dp[C + c] / xp[3 * C + c];
            v_dx[4 * C + c] = sd > 0.f ? dp[2 * C + c] / (KA_BOARD * sd) : 0.f;
        }
    }
    float sg[P16], sgy[P16], sy[P16];
#pragma unroll
    for (int e = 0; e < P16; ++e) { sg[e] = 0.f; sgy[e] = 0.f; sy[e] = 0.f; }
    // The two FC phases between the reduction and the second pass read their weights from L2 with the board's HBM traffic at a
    // standstill: as rolled loops each weight is a round trip of its own, so both loops are written load-everything-then-multiply
    // (same values, same order of additions) when the shape fits the fixed counts.
    constexpr int kW2 = 16, kW1 = 16;
    const bool w_pre = (2 * C + NTHR / H - 1) / (NTHR / H) <= kW2 && H <= kW1;
    __syncthreads();
    constexpr bool kGate = DX >= 3;          // dz not written: du is not kept
    vec16 du[kGate ? 1 : MAXSQ];
    // Gate form inside the chain (both dxc and du_up given: every launch but the one under the heads): the same arithmetic with
    // the loads of two squares -- eight 16-byte pieces per thread -- requested together.  The general loop below branches per
    // square (p < 81, the optional operands), so its four loads are one HBM round trip per square: six in a row per thread, which
    // at two boards per CU is what bounded the launch (4.1 TB/s).  A square past the board reads square 80 and adds zeros.
    bool batched = false;
    if constexpr (kGate) {
        const T* dxc = static_cast<const T*>(dxa.dxc);
        const T* dup = static_cast<const T*>(dxa.dout_up);
        batched = dxc != nullptr && dup != nullptr && H % 4 == 0;     // (H % 4: the LDS vectors below stay 16-byte aligned)
        if (batched) {
            constexpr int KB = P16 * (int)sizeof(T) < 16 ? 3 : 2;
#pragma unroll 1
            for (int i0 = 0; i0 < MAXSQ; i0 += KB) {
                vec16 ro[KB], ry[KB], r1[KB], r2[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    if (i0 + k >= MAXSQ) continue;
                    const size_t o = base + (size_t)min(slice + (i0 + k) * nsl, KA_BOARD - 1) * C;
                    ro[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(out + o));
                    ry[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(y + o));
                    r1[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(dxc + o));
                    r2[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(dup + o));
                }
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    if (i0 + k >= MAXSQ) continue;
                    const int p = slice + (i0 + k) * nsl;
                    const bool valid = p < KA_BOARD;
                    // the square's coefficients as 16-byte LDS reads, taken again per square (the offset is opaque to the
                    // compiler: held across the loop they are 48 registers, which spill)
                    int co = c0;
                    asm volatile("" : "+v"(co));
                    float gf[P16], of[P16], yf[P16], yc[P16], t[P16];
                    E::unpack(ro[k], of); E::unpack(ry[k], yf);
#pragma unroll
                    for (int h = 0; h < P16; h += 4) {           // four channels at a time: 24 coefficient registers, not 48
                        const f32x4 q0 = *reinterpret_cast<const f32x4*>(v_dx + co + h), q1 = *reinterpret_cast<const f32x4*>(v_dx + C + co + h);
                        const f32x4 q2 = *reinterpret_cast<const f32x4*>(v_dx + 2 * C + co + h), q3 = *reinterpret_cast<const f32x4*>(v_dx + 3 * C + co + h);
                        const f32x4 q4 = *reinterpret_cast<const f32x4*>(v_dx + 4 * C + co + h), q5 = *reinterpret_cast<const f32x4*>(v_mu + co + h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = of[h + e];
                            gf[h + e] = q2[e] + q4[e] * (v - q0[e]) + (v == q1[e] ? q3[e] : 0.f);
                            yc[h + e] = valid ? yf[h + e] - q5[e] : 0.f;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    E::unpack(r1[k], t);
#pragma unroll
                    for (int e = 0; e < P16; ++e) gf[e] += t[e];
                    E::unpack(r2[k], t);
#pragma unroll
                    for (int e = 0; e < P16; ++e) gf[e] += t[e];
                    const vec16 g = E::pack(gf);                 // (dx rounded to T before it is used, as in the general loop)
                    E::unpack(g, gf);
#pragma unroll
                    for (int e = 0; e < P16; ++e) {
                        gf[e] = (valid && of[e] > 0.f) ? gf[e] : 0.f;
                        sg[e] += gf[e];
                        sgy[e] = fmaf(gf[e], yc[e], sgy[e]);
                        sy[e] += yc[e];
                    }
                    if (valid) __builtin_nontemporal_store(E::pack(gf), reinterpret_cast<vec16*>(static_cast<T*>(dxa.dx) + base + (size_t)p * C));
                }
                __builtin_amdgcn_sched_barrier(0);               // (the next batch's loads hoisted above this one's arithmetic spill)
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXSQ; ++i) {
        if (kGate && batched) break;
        const int p = slice + i * nsl;
        du[kGate ? 0 : i] = vec16{};
        if (p < KA_BOARD) {
            vec16 g;
            const vec16 o = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(out + base + (size_t)p * C));
            const vec16 yv = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(y + base + (size_t)p * C));
            float gf[P16], of[P16], yf[P16];
            E::unpack(o, of);
            if constexpr (DX) {
                // dx = pooling gradients + conv1 data gradient + the residual branch of the block above (block_dx16_kernel, term for term)
                const T* dxc = static_cast<const T*>(dxa.dxc);
                const T* dup = static_cast<const T*>(dxa.dout_up);
                const T* oup = static_cast<const T*>(dxa.out_up);
                vec16 t1 = vec16{}, t2 = vec16{}, t3 = vec16{};
                if (dxc) t1 = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(dxc + base + (size_t)p * C));
                if (dup) {
                    t2 = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(dup + base + (size_t)p * C));
                    if constexpr (DX < 2) t3 = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(oup + base + (size_t)p * C));
                }
#pragma unroll
                for (int e = 0; e < P16; ++e) {
                    const float v = of[e];
                    gf[e] = v_dx[2 * C + c0 + e] + v_dx[4 * C + c0 + e] * (v - v_dx[c0 + e]) + (v == v_dx[C + c0 + e] ? v_dx[3 * C + c0 + e] : 0.f);
                }
                if (dxc) {
                    float t[P16];
                    E::unpack(t1, t);
#pragma unroll
                    for (int e = 0; e < P16; ++e) gf[e] += t[e];
                }
                if (dup) {
                    float t[P16], o2[P16];
                    E::unpack(t2, t); E::unpack(t3, o2);
                    // (a masked dout_up holds +0 where out_up <= 0: adding it as it is gives the same sum)
#pragma unroll
                    for (int e = 0; e < P16; ++e) gf[e] += (DX >= 2 || o2[e] > 0.f) ? t[e] : 0.f;
                }
                g = E::pack(gf);
                if constexpr (DX < 2) __builtin_nontemporal_store(g, reinterpret_cast<vec16*>(static_cast<T*>(dxa.dx) + base + (size_t)p * C));
            } else {
                g = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(dout + base + (size_t)p * C));
            }
            E::unpack(g, gf); E::unpack(yv, yf);
#pragma unroll
            for (int e = 0; e < P16; ++e) {
                const float yc = yf[e] - v_mu[c0 + e];
                gf[e] = of[e] > 0.f ? gf[e] : 0.f;
                sg[e] += gf[e];
                sgy[e] = fmaf(gf[e], yc, sgy[e]);      // (explicit: both instantiations must round alike)
                sy[e] += yc;
            }
            du[kGate ? 0 : i] = E::pack(gf);             // exact: a masked copy of dout
            if constexpr (DX >= 2)
                __builtin_nontemporal_store(du[kGate ? 0 : i], reinterpret_cast<vec16*>(static_cast<T*>(dxa.dx) + base + (size_t)p * C));
        }
    }
    for (int off = groups; off < 64; off <<= 1) {             // lanes cg, cg + groups, ... of a wave hold the same channels
#pragma unroll
        for (int e = 0; e < P16; ++e) {
            sg[e] += __shfl_xor(sg[e], off); sgy[e] += __shfl_xor(sgy[e], off); sy[e] += __shfl_xor(sy[e], off);
        }
    }
    if (groups >= 64 || (tid & 63) < groups) {
        const int row = tid / rowthr;
#pragma unroll
        for (int e = 0; e < P16; ++e) {
            red_g[row * C + c0 + e] = sg[e]; red_gy[row * C + c0 + e] = sgy[e]; red_y[row * C + c0 + e] = sy[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NTHR) {
        float tg = 0.f, tgy = 0.f, ty = 0.f;
        for (int s = 0; s < nrow; ++s) { tg += red_g[s * C + c]; tgy += red_gy[s * C + c]; ty += red_y[s * C + c]; }
        red_g[c] = tg; red_gy[c] = tgy; red_y[c] = ty;        // row 0 now holds the totals (read again by this thread only)
        const float scl = scale[c];
        const float t1 = fmaf(scl, tgy, fmaf(mean[c], scl, shift[c]) * tg);     // sum_p du*z
        const float sgm = sigmoidf_(se[(size_t)b * 2 * C + c]);
        const float d1 = t1 * sgm * (1.f - sgm);
        v_gate[c] = sgm;
        v_dse[c] = d1; v_dse[C + c] = tg;
        dse_out[(size_t)b * 2 * C + c] = d1; dse_out[(size_t)b * 2 * C + C + c] = tg;
    }
    __syncthreads();
    {   // dh[j] = sum_k dse[k] W2[k][j]: thread (j, part) sums every parts-th k
        const int parts = NTHR / H, j = tid % H, part = tid / H;
        float a = 0.f;
        if (w_pre) {
            float w2r[kW2];
#pragma unroll
            for (int u = 0; u < kW2; ++u) { const int k = part + u * parts; w2r[u] = k < 2 * C ? W2[(size_t)k * H + j] : 0.f; }
#pragma unroll
            for (int u = 0; u < kW2; ++u) { const int k = part + u * parts; if (k < 2 * C) a = fmaf(v_dse[k], w2r[u], a); }
        } else {
            for (int k = part; k < 2 * C; k += parts) a = fmaf(v_dse[k], W2[(size_t)k * H + j], a);
        }
        v_part[tid] = a;
    }
    __syncthreads();
    if (tid < H) {
        const int parts = NTHR / H;
        float a = 0.f;
        for (int q = 0; q < parts; ++q) a += v_part[q * H + tid];
        a = se1[(size_t)b * H + tid] > 0.f ? a : 0.f;
        v_dh[tid] = a;
        dh_out[(size_t)b * H + tid] = a;
    }
    __syncthreads();
    for (int c = tid; c < C; c += NTHR) {
        float a = 0.f;
        if (w_pre) {
            float w1r[kW1];
#pragma unroll
            for (int j = 0; j < kW1; ++j) w1r[j] = j < H ? W1[(size_t)j * C + c] : 0.f;
#pragma unroll
            for (int j = 0; j < kW1; ++j) if (j < H) a = fmaf(v_dh[j], w1r[j], a);
        } else {
            for (int j = 0; j < H; ++j) a = fmaf(v_dh[j], W1[(size_t)j * C + c], a);
        }
        v_dsq[c] = a;
        const float gate = v_gate[c], add = a / KA_BOARD;
        s1p[(size_t)b * C + c] = fmaf(gate, red_g[c], KA_BOARD * add);
        s2p[(size_t)b * C + c] = invstd[c] * fmaf(gate, red_gy[c], add * red_y[c]);
        if constexpr (kGate) { dxa.gate_out[(size_t)b * C + c] = gate; dxa.add_out[(size_t)b * C + c] = add; }
    }
    if constexpr (kGate) return;
    __syncthreads();
    float gate[P16], add[P16];
#pragma unroll
    for (int e = 0; e < P16; ++e) { gate[e] = v_gate[c0 + e]; add[e] = v_dsq[c0 + e] / KA_BOARD; }
#pragma unroll
    for (int i = 0; i < MAXSQ; ++i) {
        const int p = slice + i * nsl;
        if (p < KA_BOARD) {
            float df[P16];
            E::unpack(du[i], df);
#pragma unroll
            for (int e = 0; e < P16; ++e) df[e] = fmaf(df[e], gate[e], add[e]);
            __builtin_nontemporal_store(E::pack(df), reinterpret_cast<vec16*>(dz + base + (size_t)p * C));
        }
    }
}

// ---------------------------------------------------------------- ReLU + BN backward, reduce pass
// da = dh * [scale*y+shift > 0]  (written), partials s1 = sum da, s2 = sum da*yhat
template <typename T>
__global__ __launch_bounds__(kThreads) void relu_bn_bwd_reduce_kernel(
    const T* __restrict__ dh, const T* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    T* __restrict__ da, float* __restrict__ s1p, float* __restrict__ s2p, int C) {
    extern __shared__ float red[];
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
        if (m.active) {
            const f32x2 sc = {scale[c], scale[c + 1]}, sh = {shift[c], shift[c + 1]};
            const f32x2 mu = {mean[c], mean[c + 1]}, is = {invstd[c], invstd[c + 1]};
            for (int p = m.slice; p < KA_BOARD; p += m.ph) {
                const size_t off = base + (size_t)p * C + c;
                const f32x2 g = ld2(dh + off), yy = ld2(y + off);
                f32x2 d = {(yy[0] * sc[0] + sh[0] > 0.f) ? g[0] : 0.f, (yy[1] * sc[1] + sh[1] > 0.f) ? g[1] : 0.f};
                st2(da + off, d);
                a1[0] += d[0]; a1[1] += d[1];
                a2[0] += d[0] * ((yy[0] - mu[0]) * is[0]); a2[1] += d[1] * ((yy[1] - mu[1]) * is[1]);
            }
        }
        const f32x2 t1 = combine_sum(red, m, a1);
        const f32x2 t2 = combine_sum(red, m, a2);
        if (m.active && m.slice == 0) {
            s1p[(size_t)b * C + c] = t1[0]; s1p[(size_t)b * C + c + 1] = t1[1];
            s2p[(size_t)b * C + c] = t2[0]; s2p[(size_t)b * C + c + 1] = t2[1];
        }
    }
}

// ---------------------------------------------------------------- block input gradient
// dx = [dxc] + [dout*[out>0]] + pool_bwd(dpool; x), single streaming pass using the statistics the forward tail
// saved (xpool[b] = [mean|max|std|ties]):
//   mean: dpool_mean/81 ; max: dpool_max/ties on every square equal to the max ;
//   std : dpool_std*(x-mean)/(81*std), 0 where std == 0
template <typename T>
__global__ __launch_bounds__(kThreads) void block_dx_kernel(
    const T* __restrict__ dxc, const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x,
    const float* __restrict__ xpool, const float* __restrict__ dpool, T* __restrict__ dx, int C) {
    const BoardMap m(C);
    const int b = blockIdx.x;
    const size_t base = (size_t)b * KA_BOARD * C;
    if (!m.active) return;
    for (int cb = 0; cb < (C >> 1); cb += m.cpw) {
        const int c = (cb + m.cp) * 2;
        const float* xp = xpool + (size_t)b * 4 * C;
        const float* dp = dpool + (size_t)b * 3 * C;
        f32x2 mean, mxx, gm, gx, gs;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            mean[e] = xp[c + e]; mxx[e] = xp[C + c + e];
            const float sd = xp[2 * C + c + e];
            gm[e] = dp[c + e] / KA_BOARD;
            gx[e] = dp[C + c + e] / xp[3 * C + c + e];
            gs[e] = sd > 0.f ? dp[2 * C + c + e] / (KA_BOARD * sd) : 0.f;
        }
        for (int p = m.slice; p < KA_BOARD; p += m.ph) {
            const size_t off = base + (size_t)p * C + c;
            const f32x2 v = ld2(x + off);
            f32x2 g = {gm[0] + gs[0] * (v[0] - mean[0]) + (v[0] == mxx[0] ? gx[0] : 0.f),
                       gm[1] + gs[1] * (v[1] - mean[1]) + (v[1] == mxx[1] ? gx[1] : 0.f)};
            if (dxc) { const f32x2 t = ld2(dxc + off); g[0] += t[0]; g[1] += t[1]; }
            if (dout) {
                const f32x2 t = ld2(dout + off), o = ld2(out + off);
                g[0] += o[0] > 0.f ? t[0] : 0.f; g[1] += o[1] > 0.f ? t[1] : 0.f;
            }
            st2(dx + off, g);
        }
    }
}

// ---------------------------------------------------------------- 16-byte-per-lane forms of the two kernels above
// A thread owns one 16-byte channel piece (8 bf16 / 4 f32 channels) of every nsl-th square: 4x fewer memory
// instructions than the channel-pair form, and the forward tail keeps its squares in registers so that the pooled
// statistics are an exact two-pass computation (sum and max first, then squared deviations and ties) with two LDS
// combines instead of a serial Welford chain with a division per element.
// KB > 0: the loads of KB squares are requested together (addresses clamped to the board, a square past it adds nothing).  The
// per-square loop (KB = 0) branches on p < 81 and on the optional operands, so every square of a thread is an HBM round trip
// of its own: six in a row.  KB = 6 takes all of a thread's squares in one round trip at four waves per SIMD.
// SeArgs (sea.W1 != NULL): the squeeze-excite FC chain of the board runs HERE instead of as a launch of its own in front of this
// one (ka_block_tail_fwd_se): z = scale * (bsum[b] / 81) + shift (the BatchNorm'd channel means of y: se_resnet.py:83),
// h = relu(W1 z + b1), se = W2 h + b2 -- 2 C + H dot products per board, their weights (48 KB at C = 256, H = 16) requested at
// the kernel's first instruction; z / h / se are written out for the backward exactly as ka_fc_chain leaves them.
struct SeArgs {
    const float* bsum;                       // (B, C) per-board sums of y (the conv's epilogue output)
    const float* W1; const float* b1;        // (H, C), (H)
    const float* W2; const float* b2;        // (2C, H), (2C)
    float* sqz_out; float* se1_out; float* se_out;   // (B, C), (B, H), (B, 2C)
    int H;
};
template <typename T, int MAXSQ, int NTHR, int KB = 0, bool SEIN = false>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(((MAXSQ <= 6 || sizeof(T) == 4) && KB < 6) ? 6 : 4))) void block_tail_fwd16_kernel(
    const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ se, const T* __restrict__ res, T* __restrict__ out, float* __restrict__ pool, int C, SeArgs sea) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int P16 = E::kPer16;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int groups = C / P16, nsl = NTHR / groups;
    const int cg = tid % groups, slice = tid / groups, c0 = cg * P16;
    // partial results: the slices that share a wave are combined by lane exchanges first, one LDS row per wave (per
    // slice when a slice is wider than a wave): a quarter of the LDS of a row per slice, so more boards fit a CU
    const int rowthr = groups > 64 ? groups : 64, nrow = NTHR / rowthr, prow = tid / rowthr;
    const bool writer = groups >= 64 || (tid & 63) < groups;
    float* redA = lds;                   // [nrow][C]
    float* redB = redA + nrow * C;       // [nrow][C]
    float* redC = redB + nrow * C;       // [nrow][C]
    const size_t base = (size_t)b * KA_BOARD * C + c0;
    const bool raw = scale == nullptr;       // pooled statistics of y as it is (ka_pool_fwd): no transform, no store
    // v = (y*scale + shift)*gate + bias = y*ca + cb: the two per-channel coefficients (one sigmoid per channel) are
    // computed once per workgroup and shared through LDS (redA/redB are free until the first combine)
    if (SEIN && !raw) {
        // ---- the board's squeeze-excite chain (shapes: H <= 16 divides NTHR, C / (NTHR / H) <= 8, 2 C <= NTHR: the launcher checks)
        const int H = sea.H, parts = NTHR / H, j = tid % H, part = tid / H, kper = C / parts;
        float* zz = redC;                    // [C]            (redC is free until the second combine)
        float* sev = redC + C;               // [2C]
        float* hp = redC + 3 * C;            // [parts][H]
        float* hh = hp + NTHR;               // [H]
        constexpr int kW1 = 8, kW2 = 16;     // weights of a thread, requested before anything else is waited for
        float w1r[kW1], w2a[kW2];
#pragma unroll
        for (int u = 0; u < kW1; ++u) w1r[u] = u < kper ? sea.W1[(size_t)j * C + part * kper + u] : 0.f;
#pragma unroll
        for (int u = 0; u < kW2; ++u) w2a[u] = (u < H && tid < 2 * C) ? sea.W2[(size_t)tid * H + u] : 0.f;
        for (int c = tid; c < C; c += NTHR) {
            const float z = scale[c] * (sea.bsum[(size_t)b * C + c] * (1.f / KA_BOARD)) + shift[c];
            zz[c] = z;
            if (sea.sqz_out) sea.sqz_out[(size_t)b * C + c] = z;
        }
        __syncthreads();
        {
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < kW1; ++u) if (u < kper) a = fmaf(zz[part * kper + u], w1r[u], a);
            hp[part * H + j] = a;
        }
        __syncthreads();
        if (tid < H) {
            float a = sea.b1 ? sea.b1[tid] : 0.f;
            for (int q2 = 0; q2 < parts; ++q2) a += hp[q2 * H + tid];
            a = fmaxf(a, 0.f);
            hh[tid] = a;
            if (sea.se1_out) sea.se1_out[(size_t)b * H + tid] = a;
        }
        __syncthreads();
        if (tid < 2 * C) {
            float a = sea.b2 ? sea.b2[tid] : 0.f;
#pragma unroll
            for (int u = 0; u < kW2; ++u) if (u < H) a = fmaf(hh[u], w2a[u], a);
            sev[tid] = a;
            sea.se_out[(size_t)b * 2 * C + tid] = a;
        }
        __syncthreads();
        for (int c = tid; c < C; c += NTHR) {
            const float gt = sigmoidf_(sev[c]);
            redA[c] = scale[c] * gt;
            redB[c] = shift[c] * gt + sev[C + c];
        }
        __syncthreads();
    } else if (!raw) {
        for (int c = tid; c < C; c += NTHR) {
            const float gt = se ? sigmoidf_(se[(size_t)b * 2 * C + c]) : 1.f;
            redA[c] = scale[c] * gt;
            redB[c] = shift[c] * gt + (se ? se[(size_t)b * 2 * C + C + c] : 0.f);
        }
        __syncthreads();
    }
    float ca[P16], cb[P16], sum[P16], mx[P16];
#pragma unroll
    for (int e = 0; e < P16; ++e) {
        ca[e] = raw ? 1.f : redA[c0 + e]; cb[e] = raw ? 0.f : redB[c0 + e];
        sum[e] = 0.f; mx[e] = -INFINITY;
    }
    if (!raw) __syncthreads();               // coefficients read before redA/redB are reused
    vec16 ov[MAXSQ];
    if constexpr (KB > 0) {
        const T* resp = res ? res : y;           // (no residual: the load is wasted, not branched around)
#pragma unroll
        for (int i0 = 0; i0 < MAXSQ; i0 += KB) {
            vec16 yv[KB], rv[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (i0 + k >= MAXSQ) continue;
                const size_t o = base + (size_t)min(slice + (i0 + k) * nsl, KA_BOARD - 1) * C;
                yv[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(y + o));
                rv[k] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(resp + o));
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (i0 + k >= MAXSQ) continue;
                const int p = slice + (i0 + k) * nsl;
                const bool valid = p < KA_BOARD;
                float u[P16], r[P16];
                E::unpack(yv[k], u); E::unpack(rv[k], r);
#pragma unroll
                for (int e = 0; e < P16; ++e) {
                    float v = u[e];
                    if (!raw) {
                        v = fmaf(v, ca[e], cb[e]);
                        if (res) v += r[e];
                        v = rnd<T>(fmaxf(v, 0.f));
                    }
                    u[e] = v;
                    sum[e] += valid ? v : 0.f; mx[e] = valid ? fmaxf(mx[e], v) : mx[e];
                }
                ov[i0 + k] = valid ? E::pack(u) : vec16{};
                if (!raw && valid) __builtin_nontemporal_store(ov[i0 + k], reinterpret_cast<vec16*>(out + base + (size_t)p * C));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXSQ; ++i) {
        if (KB > 0) break;
        const int p = slice + i * nsl;
        ov[i] = vec16{};
        if (p < KA_BOARD) {
            float u[P16], r[P16];
            E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(y + base + (size_t)p * C)), u);
            if (res) E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(res + base + (size_t)p * C)), r);
#pragma unroll
            for (int e = 0; e < P16; ++e) {
                float v = u[e];
                if (!raw) {
                    v = fmaf(v, ca[e], cb[e]);
                    if (res) v += r[e];
                    v = rnd<T>(fmaxf(v, 0.f));
                }
                u[e] = v;
                sum[e] += v; mx[e] = fmaxf(mx[e], v);
            }
            ov[i] = E::pack(u);
            if (!raw) __builtin_nontemporal_store(ov[i], reinterpret_cast<vec16*>(out + base + (size_t)p * C));
        }
    }
    if (!pool) return;
    // combine 1 (one thread per channel, conflict-free): sum -> mean, max, min; totals go back through LDS row 0
    for (int off = groups; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < P16; ++e) {
            sum[e] += __shfl_xor(sum[e], off);
            mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], off));
        }
    }
    if (writer) {
#pragma unroll
        for (int e = 0; e < P16; ++e) { redA[prow * C + c0 + e] = sum[e]; redB[prow * C + c0 + e] = mx[e]; }
    }
    __syncthreads();
    float tmean = 0.f, thi = -INFINITY;
    for (int c = tid; c < C; c += NTHR) {
        float t = 0.f;
        thi = -INFINITY;
        for (int s2 = 0; s2 < nrow; ++s2) {
            t += redA[s2 * C + c];
            thi = fmaxf(thi, redB[s2 * C + c]);
        }
        tmean = t / KA_BOARD;
    }
    __syncthreads();
    for (int c = tid; c < C; c += NTHR) { redA[c] = tmean; redB[c] = thi; }
    __syncthreads();
    float mean[P16], tcnt[P16];
#pragma unroll
    for (int e = 0; e < P16; ++e) {
        mean[e] = redA[c0 + e]; mx[e] = redB[c0 + e];
        sum[e] = 0.f; tcnt[e] = 0.f;                                // sum reused: squared deviations; tie count
    }
#pragma unroll
    for (int i = 0; i < MAXSQ; ++i) {
        const int p = slice + i * nsl;
        if (p < KA_BOARD) {
            float u[P16];
            E::unpack(ov[i], u);
#pragma unroll
            for (int e = 0; e < P16; ++e) {
                const float d = u[e] - mean[e];
                sum[e] += d * d;
                tcnt[e] += u[e] == mx[e] ? 1.f : 0.f;
            }
        }
    }
    __syncthreads();
    // combine 2: squared deviations and tie counts (rows 1.. of redA/redB; row 0 still holds mean / max)
    float* sqd = redC;                       // [nrow][C]
    float* tie = redC + nrow * C;            // [nrow][C]
    for (int off = groups; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < P16; ++e) { sum[e] += __shfl_xor(sum[e], off); tcnt[e] += __shfl_xor(tcnt[e], off); }
    }
    if (writer) {
#pragma unroll
        for (int e = 0; e < P16; ++e) { sqd[prow * C + c0 + e] = sum[e]; tie[prow * C + c0 + e] = tcnt[e]; }
    }
    __syncthreads();
    float* row = pool + (size_t)b * 4 * C;
    for (int c = tid; c < C; c += NTHR) {
        float m2 = 0.f, ties = 0.f;
        for (int s2 = 0; s2 < nrow; ++s2) { m2 += sqd[s2 * C + c]; ties += tie[s2 * C + c]; }
        row[c] = redA[c];
        row[C + c] = redB[c];
        row[2 * C + c] = ties == (float)KA_BOARD ? 0.f : sqrtf(m2 / KA_BOARD);   // every square at the max: a constant plane has variance exactly 0
        row[3 * C + c] = ties;
    }
}

template <typename T, int NTHR>
__global__ __launch_bounds__(NTHR) void block_dx16_kernel(
    const T* __restrict__ dxc, const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x,
    const float* __restrict__ xpool, const float* __restrict__ dpool, T* __restrict__ dx, int C) {
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int P16 = E::kPer16;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int groups = C / P16, nsl = NTHR / groups;
    const int cg = tid % groups, slice = tid / groups, c0 = cg * P16;
    const size_t base = (size_t)b * KA_BOARD * C + c0;
    // per-channel coefficients once per workgroup (divisions!), shared through LDS: [5][C]
    extern __shared__ float lds[];
    {
        const float* xp = xpool + (size_t)b * 4 * C;
        const float* dp = dpool + (size_t)b * 3 * C;
        for (int c = tid; c < C; c += NTHR) {
            const float sd = xp[2 * C + c];
            lds[c] = xp[c];
            lds[C + c] = xp[C + c];
            lds[2 * C + c] = dp[c] / KA_BOARD;
            lds[3 * C + c] = dp[C + c] / xp[3 * C + c];
            lds[4 * C + c] = sd > 0.f ? dp[2 * C + c] / (KA_BOARD * sd) : 0.f;
        }
    }
    __syncthreads();
    float mean[P16], mxx[P16], gm[P16], gx[P16], gs[P16];
#pragma unroll
    for (int e = 0; e < P16; ++e) {
        mean[e] = lds[c0 + e]; mxx[e] = lds[C + c0 + e]; gm[e] = lds[2 * C + c0 + e];
        gx[e] = lds[3 * C + c0 + e]; gs[e] = lds[4 * C + c0 + e];
    }
#pragma unroll 3
    for (int p = slice; p < KA_BOARD; p += nsl) {
        float v[P16], g[P16];
        E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(x + base + (size_t)p * C)), v);
#pragma unroll
        for (int e = 0; e < P16; ++e) g[e] = gm[e] + gs[e] * (v[e] - mean[e]) + (v[e] == mxx[e] ? gx[e] : 0.f);
        if (dxc) {
            float t[P16];
            E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(dxc + base + (size_t)p * C)), t);
#pragma unroll
            for (int e = 0; e < P16; ++e) g[e] += t[e];
        }
        if (dout) {
            float t[P16], o[P16];
            E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(dout + base + (size_t)p * C)), t);
            E::unpack(__builtin_nontemporal_load(reinterpret_cast<const vec16*>(out + base + (size_t)p * C)), o);
#pragma unroll
            for (int e = 0; e < P16; ++e) g[e] += o[e] > 0.f ? t[e] : 0.f;
        }
        __builtin_nontemporal_store(E::pack(g), reinterpret_cast<vec16*>(dx + base + (size_t)p * C));
    }
}

// workgroup size / squares per thread of the 16-byte-per-lane board kernels (0 = shape not covered)
static int board16_plan(int C, int dtype, int* nthr) {
    const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4;
    if (C <= 0 || C % p16 != 0) return 0;
    const int groups = C / p16;
    const int nt = groups >= 32 ? 512 : 256;
    if (groups > nt || nt % groups != 0) return 0;
    *nthr = nt;
    return (KA_BOARD + nt / groups - 1) / (nt / groups);
}

template <typename T> size_t red_bytes(int C) {
    const int pairs = C >> 1, cpw = pairs < 128 ? pairs : 128;
    int ph = kThreads / cpw; if (ph > KA_BOARD) ph = KA_BOARD;
    return (size_t)ph * cpw * 9 * sizeof(float);      // PoolStat merge needs 9 floats per (slice, pair)
}

}  // namespace

#define KA_BOARD_CHECK(name) \
    KA_REQUIRE(B > 0 && C >= 2 && C % 2 == 0 && (C / 2 >= 128 ? (C / 2) % 128 == 0 : 256 / (C / 2) >= 2), \
               name ": unsupported channel count %d", C)
#define KA_DISPATCH_T(dtype, CALL) \
    do { if ((dtype) == KA_DTYPE_BF16) { typedef bf16_t T; CALL; } \
         else if ((dtype) == KA_DTYPE_F32) { typedef float T; CALL; } \
         else { ka_set_error("unknown dtype %d", (dtype)); return KA_ERR_ARG; } } while (0)

extern "C" int ka_obs_to_nhwc(const float* obs, const long long* idx, void* out, int B, int Cobs, int Cpad, int dtype,
                              void* stream) {
    KA_REQUIRE(obs && out && B > 0 && Cpad >= Cobs, "obs_to_nhwc: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(obs_to_nhwc_kernel<T>, dim3(B), dim3(256), 0, st, obs, idx, (T*)out, B, Cobs, Cpad));
    return ka_check_launch("obs_to_nhwc");
}

extern "C" int ka_nhwc_to_nchw(const void* in, float* out, int B, int C, int dtype, void* stream) {
    KA_REQUIRE(in && out && B > 0, "nhwc_to_nchw: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)in, out, B, C));
    return ka_check_launch("nhwc_to_nchw");
}

static int colsum2(const float* A, int rowsA, const float* Bp, int rowsB, int C, double* sums, double* part,
                   hipStream_t st, const char* what) {
    hipLaunchKernelGGL(colsum2_stage1_kernel, dim3((C + 63) / 64, kRedSlices), dim3(256), 0, st, A, rowsA, Bp, rowsB, C, part);
    if (sums)   // NULL: leave the partials in `part` for ka_bn_coeffs_parts / ka_bn_bwd_coeffs_parts
        hipLaunchKernelGGL(colsum2_stage2_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, st, part, sums, 2 * C);
    return ka_check_launch(what);
}

// part: workspace of ka_reduce_workspace_doubles(C) doubles
extern "C" int ka_reduce_workspace_doubles(int C) { return kRedSlices * 2 * C; }

extern "C" int ka_bn_reduce(const float* bsum, int B, const float* sqpart, int R, int C, double* sums, double* part,
                            void* stream) {
    KA_REQUIRE(bsum && sqpart && part, "bn_reduce: null tensor");
    return colsum2(bsum, B, sqpart, R, C, sums, part, static_cast<hipStream_t>(stream), "bn_reduce");
}

extern "C" int ka_pair_reduce(const float* p1, const float* p2, int B, int C, double* sums, double* part, void* stream) {
    KA_REQUIRE(p1 && p2 && part, "pair_reduce: null tensor");
    return colsum2(p1, B, p2, B, C, sums, part, static_cast<hipStream_t>(stream), "pair_reduce");
}

// SyncBatchNorm: sums[0:2C] as ka_bn_reduce / ka_pair_reduce, sums[2C] = count (this rank's element count); the caller
// all-reduces the 2C+1 doubles and hands sums + 2C to the coefficient kernels as count_dev.  local_copy (optional,
// [2C+1]) receives the same values and stays un-reduced.
extern "C" int ka_sync_reduce(const float* p1, int rows1, const float* p2, int rows2, int C, double count, double* sums,
                              double* local_copy, double* part, void* stream) {
    KA_REQUIRE(p1 && p2 && sums && part && count > 0, "sync_reduce: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum2_stage1_kernel, dim3((C + 63) / 64, kRedSlices), dim3(256), 0, st, p1, rows1, p2, rows2, C, part);
    hipLaunchKernelGGL(colsum2_stage2_sync_kernel, dim3((2 * C + 1 + 127) / 128), dim3(128), 0, st, part, sums, local_copy,
                       count, 2 * C);
    return ka_check_launch("sync_reduce");
}

extern "C" int ka_bn_coeffs(const double* sums, double count, const double* count_dev, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                            float momentum, float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                            void* stream) {
    KA_REQUIRE(sums && gamma && beta && scale && shift && mean && invstd && count > 0, "bn_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), sums, 1,
                       count, count_dev, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                       shift, mean, invstd, C);
    return ka_check_launch("bn_coeffs");
}

// as ka_bn_coeffs, reading the stage-1 partials of ka_bn_reduce(sums = NULL) directly (one launch less per layer)
extern "C" int ka_bn_coeffs_parts(const double* part, double count, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                  float eps, float* scale, float* shift, float* mean, float* invstd, int C, void* stream) {
    KA_REQUIRE(part && gamma && beta && scale && shift && mean && invstd && count > 0, "bn_coeffs_parts: bad arguments");
    hipLaunchKernelGGL(bn_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), part,
                       kRedSlices, count, nullptr, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                       eps, scale, shift, mean, invstd, C);
    return ka_check_launch("bn_coeffs_parts");
}

extern "C" int ka_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 float* scale, float* shift, int C, void* stream) {
    KA_REQUIRE(gamma && beta && rm && rv && scale && shift, "bn_eval_coeffs: null tensor");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
                       gamma, beta, rm, rv, eps, scale, shift, C);
    return ka_check_launch("bn_eval_coeffs");
}

extern "C" int ka_bn_eval_coeffs_multi(const long long* table, int n, int max_c, void* stream) {
    KA_REQUIRE(table && n > 0 && max_c > 0, "bn_eval_coeffs_multi: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_multi_kernel, dim3((max_c + 127) / 128, n), dim3(128), 0,
                       static_cast<hipStream_t>(stream), table);
    return ka_check_launch("bn_eval_coeffs_multi");
}

extern "C" int ka_bn_bwd_coeffs(const double* sums_local, const double* sums_global, double count,
                                const double* count_dev, const float* gamma, const float* mean, const float* invstd,
                                float* dgamma, float* dbeta, float* k, int C, int train, void* stream) {
    KA_REQUIRE(sums_local && sums_global && gamma && mean && invstd && k && count > 0, "bn_bwd_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
                       sums_local, sums_global, 1, count, count_dev, gamma, mean, invstd, dgamma, dbeta, k, C, train);
    return ka_check_launch("bn_bwd_coeffs");
}

// as ka_bn_bwd_coeffs, reading the stage-1 partials of ka_pair_reduce(sums = NULL) directly
extern "C" int ka_bn_bwd_coeffs_parts(const double* part, double count, const float* gamma, const float* mean,
                                      const float* invstd, float* dgamma, float* dbeta, float* k, int C, int train,
                                      void* stream) {
    KA_REQUIRE(part && gamma && mean && invstd && k && count > 0, "bn_bwd_coeffs_parts: bad arguments");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), part, part,
                       kRedSlices, count, nullptr, gamma, mean, invstd, dgamma, dbeta, k, C, train);
    return ka_check_launch("bn_bwd_coeffs_parts");
}

// ka_bn_reduce(sums = NULL) + ka_bn_coeffs_parts in one launch (colsum2_bn_coeffs_kernel).  counters: (C + 63) / 64 ints, zero before
// the first use (the kernel leaves them zero); part as ka_bn_reduce.  One such launch at a time per (part, counters) pair.
extern "C" int ka_bn_reduce_coeffs(const float* bsum, int B, const float* sqpart, int R, int C, double* part, int* counters,
                                   double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift,
                                   float* mean, float* invstd, void* stream) {
    KA_REQUIRE(bsum && sqpart && part && counters && gamma && beta && scale && shift && mean && invstd && count > 0,
               "bn_reduce_coeffs: bad arguments");
    hipLaunchKernelGGL(colsum2_bn_coeffs_kernel, dim3((C + 63) / 64, kRedSlices), dim3(256), 0, static_cast<hipStream_t>(stream), bsum, B,
                       sqpart, R, C, part, counters, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                       scale, shift, mean, invstd);
    return ka_check_launch("bn_reduce_coeffs");
}

// ka_pair_reduce(sums = NULL) + ka_bn_bwd_coeffs_parts in one launch (colsum2_bn_bwd_coeffs_kernel); counters / part as above
extern "C" int ka_pair_reduce_bwd_coeffs(const float* p1, const float* p2, int rows, int C, double* part, int* counters, double count,
                                         const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                         float* k, int train, void* stream) {
    KA_REQUIRE(p1 && p2 && part && counters && gamma && mean && invstd && k && count > 0, "pair_reduce_bwd_coeffs: bad arguments");
    hipLaunchKernelGGL(colsum2_bn_bwd_coeffs_kernel, dim3((C + 63) / 64, kRedSlices), dim3(256), 0, static_cast<hipStream_t>(stream), p1, p2,
                       rows, C, part, counters, count, gamma, mean, invstd, dgamma, dbeta, k, train);
    return ka_check_launch("pair_reduce_bwd_coeffs");
}

extern "C" int ka_affine_rows(const float* in, const float* a, const float* s, float mul, float* out, int B, int C,
                              void* stream) {
    KA_REQUIRE(in && a && s && out, "affine_rows: null tensor");
    const size_t n = (size_t)B * C;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(affine_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, a, s, mul,
                       out, B, C);
    return ka_check_launch("affine_rows");
}

extern "C" int ka_bn_bwd_apply(const void* dz, const void* y, const float* k, void* dy, int B, int C, int dtype,
                               void* stream) {
    KA_REQUIRE(dz && y && k && dy, "bn_bwd_apply: bad arguments");
    KA_BOARD_CHECK("bn_bwd_apply");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(B), dim3(kThreads), 0, st, (const T*)dz,
                                            (const T*)y, k, (T*)dy, C));
    return ka_check_launch("bn_bwd_apply");
}

static int tail_fwd_launch(const void* y, const float* scale, const float* shift, const float* se, const void* res, void* out,
                           float* pool, int B, int C, int dtype, const SeArgs& sea, hipStream_t st) {
    int nt = 0;
    const int nsq = ka_opt_set(KA_OPT_BOARD_PAIRS) ? 0 : board16_plan(C, dtype, &nt);
    if (nsq > 0 && nsq <= 11 && B > 0) {
        const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4;
        const int groups = C / p16, nrow = nt / (groups > 64 ? groups : 64);
        const size_t lds = (size_t)4 * nrow * C * sizeof(float);
#define KA_TAILF_LAUNCH_KB(MAXSQ, NTHR, KB_) \
        KA_DISPATCH_T(dtype, hipLaunchKernelGGL((block_tail_fwd16_kernel<T, MAXSQ, NTHR, KB_>), dim3(B), dim3(NTHR), lds, st, \
                                                (const T*)y, scale, shift, se, (const T*)res, (T*)out, pool, C, sea))
#define KA_TAILF_LAUNCH(MAXSQ, NTHR) KA_TAILF_LAUNCH_KB(MAXSQ, NTHR, 0)
#define KA_TAILF_LAUNCH_SE(MAXSQ, NTHR) \
        KA_DISPATCH_T(dtype, hipLaunchKernelGGL((block_tail_fwd16_kernel<T, MAXSQ, NTHR, 0, true>), dim3(B), dim3(NTHR), lds, st, \
                                                (const T*)y, scale, shift, se, (const T*)res, (T*)out, pool, C, sea))
        if (lds <= 64 * 1024) {
            if (sea.W1) {
                const int H = sea.H, parts = H > 0 && nt % H == 0 ? nt / H : 0;
                KA_REQUIRE(parts > 0 && C % parts == 0 && C / parts <= 8 && H <= 16 && 2 * C <= nt &&
                           3 * C + nt + H <= 2 * nrow * C, "block_tail_fwd_se: unsupported shape C=%d H=%d", C, H);
                if (nt == 512) { if (nsq <= 6) KA_TAILF_LAUNCH_SE(6, 512); else KA_TAILF_LAUNCH_SE(11, 512); }
                else           { if (nsq <= 6) KA_TAILF_LAUNCH_SE(6, 256); else KA_TAILF_LAUNCH_SE(11, 256); }
                return ka_check_launch("block_tail_fwd_se");
            }
            // KA_TAIL_FWD_KB: squares whose loads a thread requests together (512-thread, six-square shapes): 0 one at a time
            const int kb = ka_opt(KA_OPT_TAIL_FWD_KB, 0);
            if (nt == 512 && nsq <= 6 && kb > 0) {
                if (kb == 2) KA_TAILF_LAUNCH_KB(6, 512, 2); else if (kb == 3) KA_TAILF_LAUNCH_KB(6, 512, 3); else KA_TAILF_LAUNCH_KB(6, 512, 6);
                return ka_check_launch("block_tail_fwd");
            }
            if (nt == 512) { if (nsq <= 6) KA_TAILF_LAUNCH(6, 512); else KA_TAILF_LAUNCH(11, 512); }
            else           { if (nsq <= 6) KA_TAILF_LAUNCH(6, 256); else KA_TAILF_LAUNCH(11, 256); }
            return ka_check_launch("block_tail_fwd");
        }
#undef KA_TAILF_LAUNCH
#undef KA_TAILF_LAUNCH_KB
#undef KA_TAILF_LAUNCH_SE
    }
    KA_REQUIRE(!sea.W1, "block_tail_fwd_se: unsupported shape C=%d (ka_block_tail_fwd_se_supported)", C);
    KA_BOARD_CHECK("block_tail_fwd");
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(block_tail_fwd_kernel<T>, dim3(B), dim3(kThreads), red_bytes<T>(C), st,
                                            (const T*)y, scale, shift, se, (const T*)res, (T*)out, pool, C));
    return ka_check_launch("block_tail_fwd");
}

extern "C" int ka_block_tail_fwd(const void* y, const float* scale, const float* shift, const float* se,
                                 const void* res, void* out, float* pool, int B, int C, int dtype, void* stream) {
    KA_REQUIRE(y && scale && shift && out, "block_tail_fwd: null tensor");
    return tail_fwd_launch(y, scale, shift, se, res, out, pool, B, C, dtype, SeArgs{}, static_cast<hipStream_t>(stream));
}

// 1 when ka_block_tail_fwd_se covers (C, H, dtype): the 16-byte single-pass tail with the board's SE chain inside
extern "C" int ka_block_tail_fwd_se_supported(int C, int H, int dtype) {
    int nt = 0;
    if (ka_opt_set(KA_OPT_BOARD_PAIRS)) return 0;
    const int nsq = board16_plan(C, dtype, &nt);
    if (nsq <= 0 || nsq > 11 || H <= 0 || H > 16 || nt % H != 0) return 0;
    const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4, groups = C / p16, nrow = nt / (groups > 64 ? groups : 64), parts = nt / H;
    return C % parts == 0 && C / parts <= 8 && 2 * C <= nt && 3 * C + nt + H <= 2 * nrow * C &&
           (size_t)4 * nrow * C * sizeof(float) <= 64 * 1024;
}

// ka_block_tail_fwd with the squeeze-excite FC chain of se_resnet.py:83-86 inside: se = W2 relu(W1 z + b1) + b2 with
// z = scale * (bsum / 81) + shift (what ka_fc_chain computes from the same operands in a launch of its own); sqz_out (B, C),
// se1_out (B, H) and se_out (B, 2C) receive z, the hidden row and se for the backward.  Shapes: ka_block_tail_fwd_se_supported.
extern "C" int ka_block_tail_fwd_se(const void* y, const float* scale, const float* shift, const float* bsum, const float* W1,
                                    const float* b1, const float* W2, const float* b2, const void* res, void* out, float* pool,
                                    float* sqz_out, float* se1_out, float* se_out, int B, int C, int H, int dtype, void* stream) {
    KA_REQUIRE(y && scale && shift && out && bsum && W1 && W2 && se_out, "block_tail_fwd_se: null tensor");
    KA_REQUIRE(ka_block_tail_fwd_se_supported(C, H, dtype), "block_tail_fwd_se: unsupported shape C=%d H=%d", C, H);
    const SeArgs sea{bsum, W1, b1, W2, b2, sqz_out, se1_out, se_out, H};
    return tail_fwd_launch(y, scale, shift, nullptr, res, out, pool, B, C, dtype, sea, static_cast<hipStream_t>(stream));
}

extern "C" int ka_pool_fwd(const void* x, float* pool, int B, int C, int dtype, void* stream) {
    KA_REQUIRE(x && pool, "pool_fwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nt = 0;
    const int nsq = ka_opt_set(KA_OPT_BOARD_PAIRS) ? 0 : board16_plan(C, dtype, &nt);
    if (nsq > 0 && nsq <= 11 && B > 0) {
        const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4;
        const int groups = C / p16, nrow = nt / (groups > 64 ? groups : 64);
        const size_t lds = (size_t)4 * nrow * C * sizeof(float);
#define KA_POOL_LAUNCH(MAXSQ, NTHR) \
        KA_DISPATCH_T(dtype, hipLaunchKernelGGL((block_tail_fwd16_kernel<T, MAXSQ, NTHR>), dim3(B), dim3(NTHR), lds, st, \
                                                (const T*)x, nullptr, nullptr, nullptr, nullptr, (T*)nullptr, pool, C, SeArgs{}))
        if (lds <= 64 * 1024) {
            if (nt == 512) { if (nsq <= 6) KA_POOL_LAUNCH(6, 512); else KA_POOL_LAUNCH(11, 512); }
            else           { if (nsq <= 6) KA_POOL_LAUNCH(6, 256); else KA_POOL_LAUNCH(11, 256); }
            return ka_check_launch("pool_fwd");
        }
#undef KA_POOL_LAUNCH
    }
    KA_BOARD_CHECK("pool_fwd");
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(pool_fwd_kernel<T>, dim3(B), dim3(kThreads), red_bytes<T>(C), st,
                                            (const T*)x, pool, C));
    return ka_check_launch("pool_fwd");
}

extern "C" int ka_tail_bwd_reduce(const void* dout, const void* out, const void* y, const float* scale,
                                  const float* shift, const float* se, float* dse, int B, int C, int dtype,
                                  void* stream) {
    KA_REQUIRE(dout && out && y && scale && shift && se && dse, "tail_bwd_reduce: null tensor");
    KA_BOARD_CHECK("tail_bwd_reduce");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(tail_bwd_reduce_kernel<T>, dim3(B), dim3(kThreads), red_bytes<T>(C), st,
                                            (const T*)dout, (const T*)out, (const T*)y, scale, shift, se, dse, C));
    return ka_check_launch("tail_bwd_reduce");
}

extern "C" int ka_tail_bwd_dz(const void* dout, const void* out, const void* y, const float* se, const float* dsq,
                              const float* mean, const float* invstd, void* dz, float* s1p, float* s2p, int B, int C,
                              int dtype, void* stream) {
    KA_REQUIRE(dout && out && y && se && dsq && mean && invstd && dz && s1p && s2p, "tail_bwd_dz: null tensor");
    KA_BOARD_CHECK("tail_bwd_dz");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(tail_bwd_dz_kernel<T>, dim3(B), dim3(kThreads), red_bytes<T>(C), st,
                                            (const T*)dout, (const T*)out, (const T*)y, se, dsq, mean, invstd, (T*)dz,
                                            s1p, s2p, C));
    return ka_check_launch("tail_bwd_dz");
}

// Single-pass tail backward: workgroup size and squares per thread.  A thread keeps its squares of du and y in
// registers between the two phases, so the workgroup is sized (512 threads when the board has >= 32 channel pieces) to
// leave <= 11 squares per thread; shapes that would need more, or hidden sizes that do not divide the workgroup,
// fall back to the two-kernel path.
static int tail_fused_plan(int C, int H, int dtype, int* nthr) {
    const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4;
    if (C <= 0 || H <= 0 || C % p16 != 0) return 0;
    const int groups = C / p16;
    const int nt = groups >= 32 ? 512 : 256;
    if (groups > nt || nt % groups != 0 || H > nt || nt % H != 0) return 0;
    const int nsl = nt / groups;
    *nthr = nt;
    return (KA_BOARD + nsl - 1) / nsl;
}

extern "C" int ka_tail_bwd_fused_supported(int C, int H, int dtype) {
    int nt = 0;
    const int n = tail_fused_plan(C, H, dtype, &nt);
    return n > 0 && n <= 11;
}

// the fused block-boundary launch: shapes whose board tile fits 128 registers with the pooling coefficients (6 squares per thread;
// every fp32 shape the single-pass tail covers)
extern "C" int ka_block_dx_tail_bwd_supported(int C, int H, int dtype) {
    int nt = 0;
    const int n = tail_fused_plan(C, H, dtype, &nt);
    return n > 0 && (dtype == KA_DTYPE_F32 ? n <= 11 : n <= 6);
}

static int tail_bwd_launch(const void* dout, const void* out, const void* y, const float* scale, const float* shift,
                           const float* se, const float* se1, const float* W2, const float* W1, const float* mean,
                           const float* invstd, void* dz, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                           int H, int dtype, const DxArgs* dxp, hipStream_t st) {
    int nt = 0;
    const int nsq = tail_fused_plan(C, H, dtype, &nt);
    const int p16 = dtype == KA_DTYPE_BF16 ? 8 : 4;
    const int groups = C / p16, nrow = nt / (groups > 64 ? groups : 64);
    const size_t lds = ((size_t)3 * nrow * C + 2 * C + nt + H + 3 * C + (dxp ? 5 * C : 0)) * sizeof(float);
    KA_REQUIRE(lds <= 64 * 1024, "tail_bwd_fused: LDS footprint %zu B", lds);
    const DxArgs dxa = dxp ? *dxp : DxArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
#define KA_TAIL_LAUNCH(MAXSQ, NTHR, DX_) \
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL((tail_bwd_fused_kernel<T, MAXSQ, NTHR, DX_>), dim3(B), dim3(NTHR), lds, st, \
                                            (const T*)dout, (const T*)out, (const T*)y, scale, shift, se, se1, W2, W1, mean, \
                                            invstd, (T*)dz, dse, dh, s1p, s2p, C, H, dxa))
#define KA_TAIL_PICK(DX_) \
    if (nt == 512) { if (nsq <= 6) KA_TAIL_LAUNCH(6, 512, DX_); else KA_TAIL_LAUNCH(11, 512, DX_); } \
    else           { if (nsq <= 6) KA_TAIL_LAUNCH(6, 256, DX_); else KA_TAIL_LAUNCH(11, 256, DX_); }
    if (dxp && dxp->du_io == 2 && ka_opt(KA_OPT_TAIL_GATE_P4, 1) != 0 && dtype == KA_DTYPE_BF16 && nt == 512 && C % 4 == 0 && C / 4 <= 512 && 512 % (C / 4) == 0 &&
        (KA_BOARD + 512 / (C / 4) - 1) / (512 / (C / 4)) <= 11 && 512 % H == 0) {
        // (same LDS footprint: nrow is 8 for 64 channel groups as for 32)
        hipLaunchKernelGGL((tail_bwd_fused_kernel<bf16_t, 11, 512, 3, ElemBf4>), dim3(B), dim3(512), lds, st,
                           (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)y, scale, shift, se, se1, W2, W1, mean,
                           invstd, (bf16_t*)dz, dse, dh, s1p, s2p, C, H, dxa);
    } else if (dxp && dxp->du_io == 2) { KA_TAIL_PICK(3) } else if (dxp && dxp->du_io) { KA_TAIL_PICK(2) } else if (dxp) { KA_TAIL_PICK(1) } else { KA_TAIL_PICK(0) }
#undef KA_TAIL_PICK
#undef KA_TAIL_LAUNCH
    return ka_check_launch(dxp ? "block_dx_tail_bwd" : "tail_bwd_fused");
}

extern "C" int ka_tail_bwd_fused(const void* dout, const void* out, const void* y, const float* scale, const float* shift,
                                 const float* se, const float* se1, const float* W2, const float* W1, const float* mean,
                                 const float* invstd, void* dz, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                                 int H, int dtype, void* stream) {
    KA_REQUIRE(dout && out && y && scale && shift && se && se1 && W2 && W1 && mean && invstd && dz && dse && dh && s1p && s2p,
               "tail_bwd_fused: null tensor");
    KA_REQUIRE(B > 0 && ka_tail_bwd_fused_supported(C, H, dtype), "tail_bwd_fused: unsupported shape C=%d H=%d", C, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return tail_bwd_launch(dout, out, y, scale, shift, se, se1, W2, W1, mean, invstd, dz, dse, dh, s1p, s2p, B, C, H, dtype, nullptr, st);
}

// The block-input gradient of the block above and this block's backward tail in one launch (see DxArgs):
//   dx  = ka_block_dx(dxc, dout_up, out_up, x, xpool, dpool)          (written: the block above's `dout` is read again by
//                                                                       the residual branch one block further down)
//   ... = ka_tail_bwd_fused(dout = dx, out = x, y, ...)
// dxc may be NULL; dout_up / out_up are both NULL for the gradient that enters the tower from the heads.
extern "C" int ka_block_dx_tail_bwd(const void* dxc, const void* dout_up, const void* out_up, const void* x, const float* xpool,
                                    const float* dpool, void* dx, const void* y, const float* scale, const float* shift,
                                    const float* se, const float* se1, const float* W2, const float* W1, const float* mean,
                                    const float* invstd, void* dz, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                                    int H, int dtype, void* stream) {
    KA_REQUIRE(x && xpool && dpool && dx && ((dout_up == nullptr) == (out_up == nullptr)), "block_dx_tail_bwd: bad block_dx arguments");
    KA_REQUIRE(y && scale && shift && se && se1 && W2 && W1 && mean && invstd && dz && dse && dh && s1p && s2p,
               "block_dx_tail_bwd: null tensor");
    KA_REQUIRE(B > 0 && ka_block_dx_tail_bwd_supported(C, H, dtype), "block_dx_tail_bwd: unsupported shape C=%d H=%d", C, H);
    DxArgs dxa{dxc, dout_up, out_up, xpool, dpool, dx, 0, nullptr, nullptr};
    return tail_bwd_launch(nullptr, x, y, scale, shift, se, se1, W2, W1, mean, invstd, dz, dse, dh, s1p, s2p, B, C, H, dtype, &dxa,
                           static_cast<hipStream_t>(stream));
}

// The same launch for a chain of block boundaries, one activation read shorter: what the residual branch of the block
// below needs of this launch's dx is only du = dx * [x > 0] -- and this launch forms exactly that for its own tail.  So
//   du_out = dx * [x > 0]  is what is WRITTEN (dx itself is not), and
//   du_up  (the du_out of the launch above, or NULL for the gradient entering from the heads) is ADDED as it is,
// which takes the block above's output out of the reads (4 activation reads + 2 writes instead of 5 + 2).  dz / dse / dh /
// s1 / s2 are bit for bit those of ka_block_dx_tail_bwd; ka_block_dx accepts a du_out as its `dout` unchanged (its mask
// by the same `out` is idempotent), which is how the chain ends at the first block.
extern "C" int ka_block_dx_tail_bwd_du(const void* dxc, const void* du_up, const void* x, const float* xpool,
                                       const float* dpool, void* du_out, const void* y, const float* scale, const float* shift,
                                       const float* se, const float* se1, const float* W2, const float* W1, const float* mean,
                                       const float* invstd, void* dz, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                                       int H, int dtype, void* stream) {
    KA_REQUIRE(x && xpool && dpool && du_out, "block_dx_tail_bwd_du: bad block_dx arguments");
    KA_REQUIRE(y && scale && shift && se && se1 && W2 && W1 && mean && invstd && dz && dse && dh && s1p && s2p,
               "block_dx_tail_bwd_du: null tensor");
    KA_REQUIRE(B > 0 && ka_block_dx_tail_bwd_supported(C, H, dtype), "block_dx_tail_bwd_du: unsupported shape C=%d H=%d", C, H);
    DxArgs dxa{dxc, du_up, nullptr, xpool, dpool, du_out, 1, nullptr, nullptr};
    return tail_bwd_launch(nullptr, x, y, scale, shift, se, se1, W2, W1, mean, invstd, dz, dse, dh, s1p, s2p, B, C, H, dtype, &dxa,
                           static_cast<hipStream_t>(stream));
}

// The chain launch without dz.  dz = du_out * gate + add (gate = sigmoid of the SE gate logit, add = dsq / 81, both per
// (board, channel)) has one reader, the conv2 data gradient, whose input transform is already an affine map per channel:
// ka_conv3x3_dgrad_fused_gated takes du_out, gate_out and add_out instead of dz.  One activation write less per block
// (4 reads + 1 write); bf16(fmaf(du_out, gate_out, add_out)) is bit for bit the dz of ka_block_dx_tail_bwd_du, and du_out /
// dse / dh / s1 / s2 are the same bits.
extern "C" int ka_block_dx_tail_bwd_du_gate(const void* dxc, const void* du_up, const void* x, const float* xpool,
                                            const float* dpool, void* du_out, const void* y, const float* scale,
                                            const float* shift, const float* se, const float* se1, const float* W2,
                                            const float* W1, const float* mean, const float* invstd, float* gate_out,
                                            float* add_out, float* dse, float* dh, float* s1p, float* s2p, int B, int C,
                                            int H, int dtype, void* stream) {
    KA_REQUIRE(x && xpool && dpool && du_out, "block_dx_tail_bwd_du_gate: bad block_dx arguments");
    KA_REQUIRE(y && scale && shift && se && se1 && W2 && W1 && mean && invstd && gate_out && add_out && dse && dh && s1p && s2p,
               "block_dx_tail_bwd_du_gate: null tensor");
    KA_REQUIRE(B > 0 && ka_block_dx_tail_bwd_supported(C, H, dtype), "block_dx_tail_bwd_du_gate: unsupported shape C=%d H=%d", C, H);
    DxArgs dxa{dxc, du_up, nullptr, xpool, dpool, du_out, 2, gate_out, add_out};
    return tail_bwd_launch(nullptr, x, y, scale, shift, se, se1, W2, W1, mean, invstd, nullptr, dse, dh, s1p, s2p, B, C, H, dtype, &dxa,
                           static_cast<hipStream_t>(stream));
}

extern "C" int ka_relu_bn_bwd_reduce(const void* dh, const void* y, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, void* da, float* s1p, float* s2p, int B,
                                     int C, int dtype, void* stream) {
    KA_REQUIRE(dh && y && scale && shift && mean && invstd && da && s1p && s2p, "relu_bn_bwd_reduce: null tensor");
    KA_BOARD_CHECK("relu_bn_bwd_reduce");
    hipStream_t st = static_cast<hipStream_t>(stream);
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(relu_bn_bwd_reduce_kernel<T>, dim3(B), dim3(kThreads), red_bytes<T>(C), st,
                                            (const T*)dh, (const T*)y, scale, shift, mean, invstd, (T*)da, s1p, s2p, C));
    return ka_check_launch("relu_bn_bwd_reduce");
}

extern "C" int ka_block_dx(const void* dxc, const void* dout, const void* out, const void* x, const float* xpool,
                           const float* dpool, void* dx, int B, int C, int dtype, void* stream) {
    KA_REQUIRE(x && xpool && dpool && dx && ((dout == nullptr) == (out == nullptr)), "block_dx: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nt = 0;
    if (B > 0 && !ka_opt_set(KA_OPT_BOARD_PAIRS) && board16_plan(C, dtype, &nt) > 0) {
        if (nt == 512) {
            KA_DISPATCH_T(dtype, hipLaunchKernelGGL((block_dx16_kernel<T, 512>), dim3(B), dim3(512), (size_t)5 * C * sizeof(float), st, (const T*)dxc,
                                                    (const T*)dout, (const T*)out, (const T*)x, xpool, dpool, (T*)dx, C));
        } else {
            KA_DISPATCH_T(dtype, hipLaunchKernelGGL((block_dx16_kernel<T, 256>), dim3(B), dim3(256), (size_t)5 * C * sizeof(float), st, (const T*)dxc,
                                                    (const T*)dout, (const T*)out, (const T*)x, xpool, dpool, (T*)dx, C));
        }
        return ka_check_launch("block_dx");
    }
    KA_BOARD_CHECK("block_dx");
    KA_DISPATCH_T(dtype, hipLaunchKernelGGL(block_dx_kernel<T>, dim3(B), dim3(kThreads), 0, st, (const T*)dxc,
                                            (const T*)dout, (const T*)out, (const T*)x, xpool, dpool, (T*)dx, C));
    return ka_check_launch("block_dx");
}
