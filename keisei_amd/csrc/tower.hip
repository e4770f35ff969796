// Eval-mode residual tower in ONE launch (rollout inference, katago_ppo.py:543-617 ->
// se_resnet.py:67-75 x num_blocks under torch.no_grad() in eval mode).
//
// In eval mode no tensor couples the boards: BatchNorm uses its running statistics (a per-channel affine), the global-pool
// bias and the squeeze-excite gate are per-board reductions.  So one 512-thread workgroup carries ONE board through all
// the blocks: the board (81 x 256 bf16) and the intermediate h live in LDS, the two 3x3 convolutions of a block are the
// same implicit GEMM as conv3x3.hip (zero-haloed 17-wide LDS image of a 128-channel chunk, 9 taps as constant LDS
// offsets, fragment-ordered weights streamed L2 -> registers four k-steps ahead, out^T accumulators so a lane owns 8
// consecutive channels of a square), and everything between them stays in registers / LDS:
//   g    = W2 relu(W1 [mean|max|std](x) + b1) + b2               (global-pool bias, a per-board matrix-vector chain)
//   h    = relu(bn1(conv1(x))) + g                                 (conv1 epilogue -> LDS)
//   z    = bn2(conv2(h))                                           (accumulators)
//   s    = V2 relu(V1 mean_p(z) + c1) + c2                         (squeeze-excite chain)
//   x'   = relu(z * sigmoid(s[:C]) + s[C:] + x)                    (-> LDS, in place) and its pooled mean / max / std
// Replaces, per call of select_actions at 128 environments, 80 conv + 80 FC-chain + 40 tail launches (4.6 ms of GPU time)
// by one launch.  bf16 activations, C = 256.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kC = 256, kKC = 128;
constexpr int kPW = 17;                                   // squares per padded board row (as conv3x3.hip)
constexpr int kImgSquares = 10 * kPW + 11;                // 181
constexpr int kImgStride = kKC * 2 + 32;                  // 288 B: conflict-free 16-lane fragment reads
constexpr int kNatStride = kC * 2;                        // natural [81][256] bf16
constexpr int kXn = 0, kHn = kXn + KA_BOARD * kNatStride, kImg = kHn + KA_BOARD * kNatStride;
constexpr int kZeroSquares = 2 * (kPW + 1) + 1;           // all-zero squares behind the image: what the padded rows (81 -> 96) read
constexpr int kVec = kImg + (kImgSquares + kZeroSquares) * kImgStride;     // float vectors
constexpr int kPooled = 0, kGbias = 3 * kC, kHid = kGbias + kC, kSeMean = kHid + 256, kSeHid = kSeMean + kC, kSeOut = kSeHid + 64;
constexpr int kVecFloats = kSeOut + 2 * kC;
constexpr int kTowerLds = kVec + kVecFloats * 4;

struct TowerBlock {              // device table row, 14 pointers (int64 each on the host side)
    const char* w1; const char* w2;                       // fragment-ordered conv weights (ka_pack_conv3x3, mode 0)
    const float *sc1, *sh1, *sc2, *sh2;                   // eval-mode BatchNorm scale / shift
    const float *gw1, *gb1, *gw2, *gb2;                   // global_fc: (G, 3C), (G), (C, G), (C)
    const float *sw1, *sb1, *sw2, *sb2;                   // se_fc1 (R, C), (R); se_fc2 (2C, R), (2C)
};
struct TowerArgs {
    const uint16_t* x_in; const float* pool_in; uint16_t* x_out; float* pool_out;
    const TowerBlock* blocks; int nblocks, B, G, R;
    int abl;                     // diagnostics only (KA_TOWER_ABL): 1 skip the global-pool chain, 2 skip the SE chain, 4 skip the image builds, 8 skip the MFMA steps
};

// pointers read from the device table are generic to the compiler: loads through them would be FLAT loads, which count on
// the LDS counter as well -- every wait for an LDS fragment would then also wait for the weight prefetch.  Cast to global.
typedef const __attribute__((address_space(1))) bf16x8* gfrag_ptr;
typedef const __attribute__((address_space(1))) float* gfloat_ptr;
__device__ __forceinline__ gfloat_ptr gf(const float* p) { return (gfloat_ptr)(p); }

__device__ __forceinline__ int img_square(int p) { return (p / 9 + 1) * kPW + (p % 9) + 1; }
__device__ __forceinline__ float row16_sum(float v) {     // over the 16 lanes that share q (lane bits 0..3)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, __shfl_xor(v, 1)); v = fminf(v, __shfl_xor(v, 2)); v = fminf(v, __shfl_xor(v, 4)); v = fminf(v, __shfl_xor(v, 8));
    return v;
}

__global__ __launch_bounds__(512) void tower_eval_kernel(TowerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vec = reinterpret_cast<float*>(smem + kVec);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    const int c0 = wave * 32 + q * 8;                     // this lane's 8 output channels
    // ---- board and its pooled statistics into LDS; the image halo is zeroed once
    for (int i = tid; i < (kImgSquares + kZeroSquares) * kImgStride / 16; i += 512) reinterpret_cast<uint4*>(smem + kImg)[i] = uint4{0, 0, 0, 0};
    for (int i = tid; i < KA_BOARD * 32; i += 512)
        reinterpret_cast<uint4*>(smem + kXn)[i] = reinterpret_cast<const uint4*>(a.x_in + (size_t)b * KA_BOARD * kC)[i];
    for (int i = tid; i < 3 * kC; i += 512) vec[kPooled + i] = a.pool_in[(size_t)b * 4 * kC + i];

    // activation-fragment row offsets of this lane (6 row tiles; rows >= 81 are never stored)
    int rowoff[6];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) {
        const int p = mt * 16 + r;
        // (padded rows read zeros: zero operands cost the matrix pipe less power, and the clock is what the power budget leaves)
        rowoff[mt] = kImg + (p < KA_BOARD ? img_square(p) : kImgSquares + kPW + 1) * kImgStride + q * 16;
    }
    f32x4 acc[6][2];

    // one 3x3 convolution: src = natural-layout input in LDS, w = fragment-ordered weights
    auto conv = [&](int src, const char* w) {
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const char* wl = w + (size_t)(wave * 2) * 1024 + lane * 16;       // + ((tap*8 + ks)*16) * 1024 per k-step, + 1024 for the 2nd tile
        auto wfrag = [&](int kc, int step, bf16x8 (&f)[2]) {              // step = tap*4 + ks4 within the chunk (clamped)
            step = min(step, 35);
            const int tap = step >> 2, ks = kc * 4 + (step & 3);
            const char* p = wl + (size_t)((tap * 8 + ks) * 16) * 1024;
            f[0] = *(gfrag_ptr)(p);
            f[1] = *(gfrag_ptr)(p + 1024);
        };
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 w0[2], w1[2], w2[2], w3[2];
            wfrag(kc, 0, w0); wfrag(kc, 1, w1); wfrag(kc, 2, w2);         // in flight across the image build
            KA_LDS_BARRIER();                                              // the image's previous readers are done; src is complete
            if (!(a.abl & 4)) for (int i = tid; i < KA_BOARD * 16; i += 512) {
                const int row = i >> 4, pc = i & 15;
                *reinterpret_cast<uint4*>(smem + kImg + img_square(row) * kImgStride + pc * 16) =
                    *reinterpret_cast<const uint4*>(smem + src + row * kNatStride + kc * 256 + pc * 16);
            }
            KA_LDS_BARRIER();
            auto toff_of = [&](int step) {
                step = min(step, 35);
                const int tap = step >> 2, ks = step & 3;
                return ((tap / 3 - 1) * kPW + (tap % 3 - 1)) * kImgStride + ks * 64;
            };
            // activation fragments are double-buffered across k-steps: the 6 LDS reads of step s + 1 are issued one in
            // front of every pair of MFMAs of step s (issue order pinned), so an MFMA never waits for the read just issued
            auto mm = [&](const bf16x8 (&wf)[2], const bf16x8 (&ac)[6], bf16x8 (&an)[6], int next_step) {
                const int toff = toff_of(next_step);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) {
                    an[mt] = *reinterpret_cast<const bf16x8*>(smem + rowoff[mt] + toff);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], ac[mt], acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], ac[mt], acc[mt][1], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            };
            bf16x8 fa[6], fb[6];
            {
                const int toff = toff_of(0);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(smem + rowoff[mt] + toff);
            }
#pragma unroll 1
            for (int tap = 0; tap < ((a.abl & 8) ? 0 : 9); ++tap) {      // weights of step s + 3 are requested before step s multiplies
                const int s0 = tap * 4;
                wfrag(kc, s0 + 3, w3); __builtin_amdgcn_sched_barrier(0); mm(w0, fa, fb, s0 + 1); __builtin_amdgcn_sched_barrier(0);
                wfrag(kc, s0 + 4, w0); __builtin_amdgcn_sched_barrier(0); mm(w1, fb, fa, s0 + 2); __builtin_amdgcn_sched_barrier(0);
                wfrag(kc, s0 + 5, w1); __builtin_amdgcn_sched_barrier(0); mm(w2, fa, fb, s0 + 3); __builtin_amdgcn_sched_barrier(0);
                wfrag(kc, s0 + 6, w2); __builtin_amdgcn_sched_barrier(0); mm(w3, fb, fa, s0 + 4); __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    KA_LDS_BARRIER();
    // Every workgroup walks the same FC weight rows, and all of them arrive at a chain at about the same moment: without a
    // per-workgroup rotation of the row order all 128+ workgroups ask the same L2 lines in the same clock (one channel serves
    // them one after the other while the others idle).  The rows are independent, so the order does not touch the results.
    const int rot1 = (int)((blockIdx.x * 8u) % (unsigned)a.G), rot2 = (int)((blockIdx.x * 16u) & (kC - 1));
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const TowerBlock tb = a.blocks[blk];
        // ---- global-pool bias: hid = relu(W1 pooled + b1), one wave per row, lanes along the 3C inputs; four rows' loads are
        // in flight together (a dependent chain of L2 round trips per row is what this phase would otherwise be)
        if (!(a.abl & 1)) {
            // 16-byte loads, eight rows (24 loads per lane) in flight together: the phase is a chain of L2 round trips
            typedef const __attribute__((address_space(1))) f32x4* gvec_ptr;
            f32x4 pv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) pv[k] = *reinterpret_cast<const f32x4*>(vec + kPooled + 4 * (lane + 64 * k));
            for (int jb = wave; jb < a.G; jb += 64) {
                f32x4 wv[8][3];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = (min(jb + 8 * u, a.G - 1) + rot1) % a.G;          // (rows rotated per workgroup, see rot1)
                    gvec_ptr wr = (gvec_ptr)(tb.gw1 + (size_t)j * 3 * kC);
#pragma unroll
                    for (int k = 0; k < 3; ++k) wv[u][k] = wr[lane + 64 * k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) t += wv[u][k][0] * pv[k][0] + wv[u][k][1] * pv[k][1] + wv[u][k][2] * pv[k][2] + wv[u][k][3] * pv[k][3];
                    t = wave_sum(t);
                    const int j = jb + 8 * u, jr = (j + rot1) % a.G;
                    if (lane == 0 && j < a.G) vec[kHid + jr] = fmaxf(t + gf(tb.gb1)[jr], 0.f);
                }
            }
        }
        KA_LDS_BARRIER();
        if (!(a.abl & 1)) {   // g[c] = W2[c] . hid + b2[c]: two threads per channel, each a contiguous half of the row (16-byte loads)
            typedef const __attribute__((address_space(1))) f32x4* gvec_ptr;
            const int c = ((tid >> 1) + rot2) & (kC - 1), half = tid & 1, n = a.G >> 1;
            gvec_ptr wr = (gvec_ptr)(tb.gw2 + (size_t)c * a.G + half * n);
            float s = 0.f;
            for (int j = 0; j < n / 4; ++j) {
                const f32x4 w4 = wr[j];
                const float* hv = vec + kHid + half * n + 4 * j;
                s += w4[0] * hv[0] + w4[1] * hv[1] + w4[2] * hv[2] + w4[3] * hv[3];
            }
            s += __shfl_xor(s, 1);
            if (half == 0) vec[kGbias + c] = s + gf(tb.gb2)[c];
        }
        // ---- conv1 and its epilogue: h = relu(bn1(y1)) + g   (the first barrier inside conv publishes g)
        conv(kXn, tb.w1);
        {
            float sc[8], sh[8], gb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = gf(tb.sc1)[c0 + e]; sh[e] = gf(tb.sh1)[c0 + e]; gb[e] = vec[kGbias + c0 + e]; }
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const int p = mt * 16 + r;
                if (p < KA_BOARD) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // y1 is rounded to bf16 first, as the stand-alone kernels store it, then transformed and rounded again
                        const float y0 = (float)(__bf16)acc[mt][0][e], y1 = (float)(__bf16)acc[mt][1][e];
                        o[e] = (__bf16)(fmaxf(fmaf(y0, sc[e], sh[e]), 0.f) + gb[e]);
                        o[4 + e] = (__bf16)(fmaxf(fmaf(y1, sc[4 + e], sh[4 + e]), 0.f) + gb[4 + e]);
                    }
                    *reinterpret_cast<bf16x8*>(smem + kHn + p * kNatStride + c0 * 2) = o;
                }
            }
        }
        // ---- conv2; z = bn2(y2) stays in the accumulators
        conv(kHn, tb.w2);
        {
            float sc[8], sh[8], sm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = gf(tb.sc2)[c0 + e]; sh[e] = gf(tb.sh2)[c0 + e]; sm[e] = 0.f; }
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const bool ok = mt * 16 + r < KA_BOARD;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (the squeeze is the mean of the fp32 conv output, as the conv epilogue of the stand-alone path takes it)
                    sm[e] += ok ? acc[mt][0][e] : 0.f; sm[4 + e] += ok ? acc[mt][1][e] : 0.f;
                    acc[mt][0][e] = fmaf((float)(__bf16)acc[mt][0][e], sc[e], sh[e]);
                    acc[mt][1][e] = fmaf((float)(__bf16)acc[mt][1][e], sc[4 + e], sh[4 + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = row16_sum(sm[e]);
                if (r == 0) vec[kSeMean + c0 + e] = fmaf(t * (1.f / KA_BOARD), sc[e], sh[e]);   // mean_p bn2(y2)
            }
        }
        KA_LDS_BARRIER();
        if (!(a.abl & 2)) {   // se hidden: one wave per row, the loads of all this wave's rows in flight together
            float mv[kC / 64];
#pragma unroll
            for (int k = 0; k < kC / 64; ++k) mv[k] = vec[kSeMean + lane + 64 * k];
            for (int jb = wave; jb < a.R; jb += 32) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(jb + 8 * u, a.R - 1);
                    gfloat_ptr wr = gf(tb.sw1) + (size_t)j * kC;
#pragma unroll
                    for (int k = 0; k < kC / 64; ++k) s[u] += wr[lane + 64 * k] * mv[k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = jb + 8 * u;
                    const float t = wave_sum(s[u]);
                    if (lane == 0 && j < a.R) vec[kSeHid + j] = fmaxf(t + gf(tb.sb1)[j], 0.f);
                }
            }
        }
        KA_LDS_BARRIER();
        if (!(a.abl & 2)) {
            gfloat_ptr wr = gf(tb.sw2) + (size_t)tid * a.R;   // thread k < 2C = 512: gate logits | shifts
            float s = gf(tb.sb2)[tid];
            if ((a.R & 3) == 0) {
                typedef const __attribute__((address_space(1))) f32x4* gvec_ptr;
                gvec_ptr w4p = (gvec_ptr)(tb.sw2 + (size_t)tid * a.R);
                for (int j = 0; j < a.R / 4; ++j) {
                    const f32x4 w4 = w4p[j];
                    const float* hv = vec + kSeHid + 4 * j;
                    s += w4[0] * hv[0] + w4[1] * hv[1] + w4[2] * hv[2] + w4[3] * hv[3];
                }
            } else {
                for (int j = 0; j < a.R; ++j) s += wr[j] * vec[kSeHid + j];
            }
            vec[kSeOut + tid] = tid < kC ? sigmoidf_(s) : s;
        }
        KA_LDS_BARRIER();
        // ---- x' = relu(z * gate + shift + x) in place, and its pooled statistics (a wave owns all squares of its channels)
        {
            float gate[8], shf[8], sum[8], mx[8], mn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gate[e] = vec[kSeOut + c0 + e]; shf[e] = vec[kSeOut + kC + c0 + e]; sum[e] = 0.f; mx[e] = -INFINITY; mn[e] = INFINITY; }
            float v[6][8];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const int p = mt * 16 + r;
                const bool ok = p < KA_BOARD;
                bf16x8 xr = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) xr = *reinterpret_cast<const bf16x8*>(smem + kXn + p * kNatStride + c0 * 2);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = e < 4 ? acc[mt][0][e] : acc[mt][1][e - 4];
                    o[e] = (__bf16)fmaxf(fmaf(z, gate[e], shf[e]) + (float)xr[e], 0.f);
                    v[mt][e] = (float)o[e];
                    if (ok) { sum[e] += v[mt][e]; mx[e] = fmaxf(mx[e], v[mt][e]); mn[e] = fminf(mn[e], v[mt][e]); }
                }
                if (ok) *reinterpret_cast<bf16x8*>(smem + kXn + p * kNatStride + c0 * 2) = o;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mean = row16_sum(sum[e]) * (1.f / KA_BOARD);
                const float hi = row16_max(mx[e]), lo = row16_min(mn[e]);
                float m2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) { const float d = v[mt][e] - mean; m2 += (mt * 16 + r < KA_BOARD) ? d * d : 0.f; }
                m2 = row16_sum(m2);
                if (r == 0) {
                    vec[kPooled + c0 + e] = mean;
                    vec[kPooled + kC + c0 + e] = hi;
                    vec[kPooled + 2 * kC + c0 + e] = hi == lo ? 0.f : sqrtf(m2 * (1.f / KA_BOARD));   // a constant plane: exactly 0
                }
            }
        }
        KA_LDS_BARRIER();
    }
    for (int i = tid; i < KA_BOARD * 32; i += 512)
        reinterpret_cast<uint4*>(a.x_out + (size_t)b * KA_BOARD * kC)[i] = reinterpret_cast<const uint4*>(smem + kXn)[i];
    for (int i = tid; i < 4 * kC; i += 512) a.pool_out[(size_t)b * 4 * kC + i] = i < 3 * kC ? vec[kPooled + i] : 0.f;
}

}  // namespace

// 1 when the one-launch tower covers this configuration (bf16 activations, 256 channels, FC widths it holds in LDS)
extern "C" int ka_tower_eval_supported(int C, int G, int R, int dtype) {
    return dtype == KA_DTYPE_BF16 && C == kC && G >= 8 && G <= 256 && G % 8 == 0 && R >= 1 && R <= 64;
}

// x_out, pool_out = the residual tower applied to x_in (B, 81, C) bf16 with pooled statistics pool_in (B, 4C); blocks =
// device table of nblocks x 14 pointers (TowerBlock).  Eval mode only (BatchNorm as scale/shift).
extern "C" int ka_tower_eval(const void* x_in, const float* pool_in, void* x_out, float* pool_out, const void* blocks,
                             int nblocks, int B, int C, int G, int R, int dtype, void* stream) {
    KA_REQUIRE(x_in && pool_in && x_out && pool_out && blocks && nblocks > 0 && B > 0, "tower_eval: bad arguments");
    KA_REQUIRE(ka_tower_eval_supported(C, G, R, dtype), "tower_eval: unsupported configuration C=%d G=%d R=%d dtype=%d", C, G, R, dtype);
    TowerArgs a{static_cast<const uint16_t*>(x_in), pool_in, static_cast<uint16_t*>(x_out), pool_out,
                static_cast<const TowerBlock*>(blocks), nblocks, B, G, R, 0};
    if (const char* e = ka_diag_env("KA_TOWER_ABL")) a.abl = atoi(e);
    static std::atomic<unsigned long long> done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&tower_eval_kernel), done, "tower_eval")) return rc;
    hipLaunchKernelGGL(tower_eval_kernel, dim3(B), dim3(512), kTowerLds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("tower_eval");
}
