// 3x3 "same" convolution over 9x9 boards as an implicit GEMM on the CDNA4 matrix cores.
//
//   out[b,p,n] = sum_{tap,c} X'[b, p+tap, c] * W[n,c,tap]      X' = optional fused transform of the input
//
// Layout: activations are NHWC (B,81,C), channel-contiguous, so every HBM access is a
// coalesced run along the channel axis.  One workgroup (4 waves) owns NB=2 whole boards and a
// slab of output channels: the boards are staged ONCE into LDS as zero-haloed 11x11 tiles
// ([padded square][channel]), after which each of the 9 taps is just a constant LDS offset
// ((dy*17+dx)*row_stride) on the MFMA A-operand read -- the input is read from HBM exactly
// once per output-channel slab.  Weights are pre-packed (pack_conv3x3_kernel) in MFMA
// B-fragment order so a wave streams them from L2 with perfectly coalesced 1 KiB loads
// straight into registers; waves split the N (output channel) dimension, so no weight goes
// through LDS.  M = 162 rows is padded to 11 tiles of 16 (v_mfma_f32_16x16x32_bf16 /
// 4x v_mfma_f32_16x16x4_f32); the f32 path is an exact fmaf chain (parity mode), the bf16
// path is the throughput mode.
//
// The epilogue fuses what the following BatchNorm / squeeze-excite need: per-board channel
// sums (= SE squeeze, and summed over boards the BN mean) and per-workgroup sums of squares.
//
// The same kernel is the data-gradient conv: pack with flip+transpose (mode 1).
//
// Reference semantics replaced: nn.Conv2d(C, C, 3, padding=1, bias=False) at
// keisei/training/models/se_resnet.py:50,52,110 and its autograd backward.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kNB = 2;                       // boards per workgroup
constexpr int kRows = kNB * KA_BOARD;        // 162 GEMM rows
constexpr int kMT = (kRows + 15) / 16;       // 11 row tiles
// LDS image of the input: zero-haloed boards, [square][channel].  A padded board row is 17 squares wide (9 + halo,
// widened so that stepping to the next board row advances the square index by 8 more than a neighbour step) and the
// second board starts 193 squares after the first; with a row stride of 32 bytes more than a multiple of 256 a
// 16-lane MFMA fragment read (16 consecutive GEMM rows x 16 bytes) then lands on 16 different 16-byte bank slots for
// every tap -- conflict-free ds_read_b128, where the natural 11-wide image costs a 2-way conflict on every read.
constexpr int kPW = 17;                      // squares per padded board row
constexpr int kBoardStride = 193;            // squares between the two boards
constexpr int kLdsSquares = kBoardStride + 11 * kPW;   // 380
__device__ __forceinline__ int lds_square(int b, int p) { return b * kBoardStride + (p / 9 + 1) * kPW + (p % 9) + 1; }

struct ConvArgs {
    const void* in;
    const void* wpack;
    void* out;
    const float* in_scale;   // [Cin] or null: x' = x*scale + shift
    const float* in_shift;
    const float* in_bias;    // [B,Cin] or null: per-board bias added after the ReLU
    float* bsum;             // [B,Cout] or null
    float* sqpart;           // [gridDim.x, Cout] or null
    int B, Cin, Cout, KC, relu;
    // fused BatchNorm-backward input (data-gradient convs): x' = in*in_scale + in_shift + in2*in_k3, optionally
    // written back to in_out (the materialised dy the weight-gradient kernel reads)
    const void* in2;
    const float* in_k3;
    void* in_out;
    // fused ReLU+BatchNorm-backward epilogue (bf16): out = acc * [ep_scale*ep_y + ep_shift > 0]; per-workgroup partial
    // sums ep_s1[wg][n] = sum out, ep_s2[wg][n] = sum out*(ep_y-ep_mean)*ep_invstd
    const void* ep_y;
    const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_invstd;
    float* ep_s1; float* ep_s2;
    int tune_stagger, tune_prio;  // experiments: s_sleep count / static priority of the second wave of every SIMD
    unsigned long long* stamps;   // diagnostic only: [workgroup][8] s_memtime at phase boundaries (null in production)
};

unsigned long long* g_stamps = nullptr;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const bf16x8& a, const bf16x8& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane (r,q) holds channels 4q..4q+3 of its row/column: MFMA i contracts channel 4q+i over q
    static __device__ __forceinline__ f32x4 run(const f32x4& a, const f32x4& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// WM = waves along M: 1 -> 4 waves, each owns all 11 row tiles of its output-channel slice;
//                     2 -> 8 waves (2 per SIMD), each owns 6 row tiles: half the accumulators per wave, and a
//                          second wave per SIMD whose MFMAs fill the first one's LDS/L2 stalls
template <typename T, int NTW, int WM>
__global__ __launch_bounds__(256 * WM) void conv3x3_kernel(ConvArgs a) {
    constexpr int kThreads = 256 * WM;
    constexpr int kMTW = (kMT + WM - 1) / WM;        // row tiles per wave
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int ESZ = E::kSize, P16 = E::kPer16, CPK = 4 * P16;   // channels per k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, mhalf = tid >> 8;
    const int mt_base = mhalf * kMTW;
    const int r = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * kNB;
    const int NT = a.Cout >> 4;
    const int nt0 = blockIdx.y * (4 * NTW) + wave * NTW;
    const int stride = a.KC * ESZ + 32;          // bytes per padded square (+32: bank spread, see kPW)
    const int cpr = a.KC * ESZ / 16;             // 16-byte pieces per square
    const int KSG = a.Cin / CPK;                 // k-steps over all input channels
    const int KS = a.KC / CPK;                   // k-steps per LDS chunk

    f32x4 acc[kMTW][NTW];
#pragma unroll
    for (int mt = 0; mt < kMTW; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int rowoff[kMTW];
#pragma unroll
    for (int mt = 0; mt < kMTW; ++mt) {
        int m = (mt_base + mt) * 16 + r;
        if (m >= kRows) m = 0;                   // dummy rows read valid LDS; never stored
        rowoff[mt] = lds_square(m / KA_BOARD, m % KA_BOARD) * stride + q * 16;
    }

    // zero the halo once (staging only ever writes interior squares)
    for (int i = tid; i < kLdsSquares * cpr; i += kThreads) {
        const int idx = i / cpr, j = i - idx * cpr;
        const int pp = idx >= kBoardStride ? idx - kBoardStride : idx, yy = pp / kPW, xx = pp - yy * kPW;
        const bool interior = yy >= 1 && yy <= 9 && xx >= 1 && xx <= 9 && pp < 11 * kPW;
        if (!interior) *reinterpret_cast<uint4*>(smem + idx * stride + j * 16) = uint4{0, 0, 0, 0};
    }

    const int sj = tid % cpr, spos0 = tid / cpr, sstep = kThreads / cpr;   // staging role of this thread
    const char* wbase = static_cast<const char*>(a.wpack) + ((size_t)nt0 * 64 + lane) * 16;
    const size_t tap_stride = (size_t)KSG * NT * 1024, ks_stride = (size_t)NT * 1024;
    const bool wave_active = nt0 < NT;

    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 0] = __builtin_amdgcn_s_memtime();
    const int nchunks = a.Cin / a.KC;
    for (int kc = 0; kc < nchunks; ++kc) {
        if (kc > 0) __syncthreads();
        // ---- stage NB boards x KC channels into the haloed LDS tile (fused input transform)
        {
            const int c0 = kc * a.KC + sj * P16;
            float sc[P16], sh[P16];
            const bool has_aff = a.in_scale != nullptr;
            if (has_aff) {
#pragma unroll
                for (int e = 0; e < P16; ++e) { sc[e] = a.in_scale[c0 + e]; sh[e] = a.in_shift[c0 + e]; }
            }
            float gbv[kNB][P16];                 // per-board bias of this thread's channel piece, fetched once
            if (a.in_bias) {
#pragma unroll
                for (int b = 0; b < kNB; ++b)
#pragma unroll
                    for (int e = 0; e < P16; ++e)
                        gbv[b][e] = (b0 + b < a.B) ? a.in_bias[(size_t)(b0 + b) * a.Cin + c0 + e] : 0.f;
            }
            if (a.in2) {
                // two-tensor transform (BatchNorm backward applied on the fly), 3 + 3 loads in flight
                float k3[P16];
#pragma unroll
                for (int e = 0; e < P16; ++e) k3[e] = a.in_k3[c0 + e];
                constexpr int kU2 = (WM == 2) ? 6 : 4;
                for (int pos0 = spos0; pos0 < kRows; pos0 += sstep * kU2) {
                    vec16 v[kU2], w[kU2];
#pragma unroll
                    for (int u = 0; u < kU2; ++u) {
                        const int pos = pos0 + u * sstep;
                        const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                        const size_t off = ((size_t)(bb * KA_BOARD + p) * a.Cin + c0) * ESZ;
                        if (pos < kRows && bb < a.B) {
                            v[u] = *reinterpret_cast<const vec16*>(static_cast<const char*>(a.in) + off);
                            w[u] = *reinterpret_cast<const vec16*>(static_cast<const char*>(a.in2) + off);
                        } else { v[u] = vec16{}; w[u] = vec16{}; }
                    }
#pragma unroll
                    for (int u = 0; u < kU2; ++u) {
                        const int pos = pos0 + u * sstep;
                        if (pos >= kRows) continue;
                        const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                        if (bb < a.B) {
                            float f[P16], g2[P16];
                            E::unpack(v[u], f);
                            E::unpack(w[u], g2);
#pragma unroll
                            for (int e = 0; e < P16; ++e) f[e] = fmaf(g2[e], k3[e], fmaf(f[e], sc[e], sh[e]));
                            v[u] = E::pack(f);
                            if (a.in_out && blockIdx.y == 0)
                                *reinterpret_cast<vec16*>(static_cast<char*>(a.in_out) +
                                                          ((size_t)(bb * KA_BOARD + p) * a.Cin + c0) * ESZ) = v[u];
                        }
                        *reinterpret_cast<vec16*>(smem + lds_square(b, p) * stride + sj * 16) = v[u];
                    }
                }
            } else {
            // loads are issued in batches of kUnr before any is consumed: the staging phase is otherwise a
            // chain of dependent HBM round trips (one per 16-byte piece per thread)
            constexpr int kUnr = (WM == 2) ? 6 : 11;
            for (int pos0 = spos0; pos0 < kRows; pos0 += sstep * kUnr) {
                vec16 v[kUnr];
#pragma unroll
                for (int u = 0; u < kUnr; ++u) {
                    const int pos = pos0 + u * sstep;
                    const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                    if (pos < kRows && bb < a.B)
                        v[u] = *reinterpret_cast<const vec16*>(static_cast<const char*>(a.in) +
                                                               ((size_t)(bb * KA_BOARD + p) * a.Cin + c0) * ESZ);
                    else
                        v[u] = vec16{};
                }
#pragma unroll
                for (int u = 0; u < kUnr; ++u) {
                    const int pos = pos0 + u * sstep;
                    if (pos >= kRows) continue;
                    const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                    if (bb < a.B && (has_aff || a.relu || a.in_bias)) {
                        float f[P16];
                        E::unpack(v[u], f);
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
                            for (int e = 0; e < P16; ++e) f[e] += (b == 0 ? gbv[0][e] : gbv[1][e]);
                        }
                        v[u] = E::pack(f);
                    }
                    *reinterpret_cast<vec16*>(smem + lds_square(b, p) * stride + sj * 16) = v[u];
                }
            }
                    }
        }
        __syncthreads();
        if (a.stamps && tid == 0 && kc == 0) a.stamps[wg_lin * 8 + 1] = __builtin_amdgcn_s_memtime();

        // ---- MFMA phase: 9 taps x KS k-steps.  Weight fragments stream from L2 straight into registers,
        // ping-ponged between two named register sets (no conditional loads, no register copies) so the
        // compiler's vmcnt bookkeeping leaves the next step's loads in flight under this step's MFMAs.
        if (wave_active) {
            if (WM == 2 && kc == 0 && __builtin_amdgcn_readfirstlane(mhalf) == 1) {
                if (a.tune_prio) __builtin_amdgcn_s_setprio(1);
                for (int z = 0; z < a.tune_stagger; ++z) __builtin_amdgcn_s_sleep(2);
            }
            const int nsteps = 9 * KS;
            const char* wchunk = wbase + (size_t)(kc * KS) * ks_stride;
            // tile indices beyond NT (partial last wave) are clamped: they load valid bytes that are never stored
            int jofs[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) jofs[j] = (min(nt0 + j, NT - 1) - nt0) * 1024;
            auto wptr = [&](int step) {
                step = min(step, nsteps - 1);
                const int tap = step / KS, ks = step - tap * KS;
                return wchunk + (size_t)tap * tap_stride + (size_t)ks * ks_stride;
            };
            auto lds_off = [&](int step) {
                const int tap = step / KS, ks = step - tap * KS;
                return ((tap / 3 - 1) * kPW + (tap % 3 - 1)) * stride + ks * 64;
            };
            // A fragments are double-buffered in registers across steps: while step s multiplies from `ac`, the
            // 11 LDS reads of step s+1 land in `an` (a whole step of MFMA time to hide LDS latency/conflicts).
            auto load_a = [&](vec16 (&av)[kMTW], int toff) {
#pragma unroll
                for (int mt = 0; mt < kMTW; ++mt) av[mt] = *reinterpret_cast<const vec16*>(smem + rowoff[mt] + toff);
            };
            auto compute = [&](const vec16 (&bw)[NTW], const vec16 (&av)[kMTW], vec16 (&anext)[kMTW], int toff_next) {
#pragma unroll
                for (int mt = 0; mt < kMTW; ++mt) {
#ifndef KA_DIAG_NO_A
                    anext[mt] = *reinterpret_cast<const vec16*>(smem + rowoff[mt] + toff_next);
#else
                    anext[mt] = av[mt];
#endif
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[mt][j] = Mma<T>::run(av[mt], bw[j], acc[mt][j]);
                }
                // pin the issue order: one LDS read ahead of every group of MFMAs (the scheduler otherwise
                // sinks all 11 reads behind the MFMA block and the next step starts by waiting for them)
#pragma unroll
                for (int mt = 0; mt < kMTW; ++mt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NTW * (sizeof(T) == 2 ? 1 : 4), 0);
                }
            };
            vec16 b0[NTW], b1[NTW], a0[kMTW], a1[kMTW];
            {
                const char* wp = wptr(0);
#pragma unroll
                for (int j = 0; j < NTW; ++j) b0[j] = *reinterpret_cast<const vec16*>(wp + jofs[j]);
            }
            load_a(a0, lds_off(0));
            for (int it = 0; it < nsteps; it += 2) {
#ifndef KA_DIAG_NO_W
                {
                    const char* wp = wptr(it + 1);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) b1[j] = *reinterpret_cast<const vec16*>(wp + jofs[j]);
                }
#else
#pragma unroll
                for (int j = 0; j < NTW; ++j) b1[j] = b0[j];
#endif
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE the MFMA block it overlaps
                compute(b0, a0, a1, lds_off(min(it + 1, nsteps - 1)));
#ifndef KA_DIAG_NO_W
                {
                    const char* wp = wptr(it + 2);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) b0[j] = *reinterpret_cast<const vec16*>(wp + jofs[j]);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < nsteps) compute(b1, a1, a0, lds_off(min(it + 2, nsteps - 1)));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: statistics from the fp32 accumulators, then the output tile
    const bool v0 = b0 < a.B, v1 = b0 + 1 < a.B;
    constexpr int BN = 4 * NTW * 16;                 // output channels of this workgroup
    constexpr int ostride = BN * 2 + 16;
    float* stat_lds = reinterpret_cast<float*>(smem + (sizeof(T) == 2 ? kMT * 16 * ostride : 0));   // [WM][BN][3]
    const bool want_stats = a.bsum || a.sqpart;
    float s0[NTW], s1[NTW], ss[NTW];
    if (wave_active && want_stats) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            s0[j] = s1[j] = ss[j] = 0.f;
#pragma unroll
            for (int mt = 0; mt < kMTW; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = (mt_base + mt) * 16 + q * 4 + i;
                    const float v = acc[mt][j][i];
                    if (m < KA_BOARD) { s0[j] += v; ss[j] += v * v; }
                    else if (m < kRows) { s1[j] += v; ss[j] += v * v; }
                }
            s0[j] += __shfl_xor(s0[j], 16); s0[j] += __shfl_xor(s0[j], 32);
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            ss[j] += __shfl_xor(ss[j], 16); ss[j] += __shfl_xor(ss[j], 32);
        }
    }
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 3] = __builtin_amdgcn_s_memtime();
    if (WM > 1 || sizeof(T) == 2) __syncthreads();   // all waves done reading the input tile
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 4] = __builtin_amdgcn_s_memtime();
    if (WM > 1 && wave_active && want_stats && mhalf == 1 && q == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            float* d = stat_lds + ((wave * NTW + j) * 16 + r) * 3;
            d[0] = s0[j]; d[1] = s1[j]; d[2] = ss[j];
        }
    }
    if constexpr (sizeof(T) == 2) {
        // transpose through LDS so HBM sees whole 16-byte pieces of contiguous rows
        if (wave_active) {
#pragma unroll
            for (int mt = 0; mt < kMTW; ++mt)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt_base + mt) * 16 + q * 4 + i;
                        if (m < kMT * 16)
                            *reinterpret_cast<uint16_t*>(smem + m * ostride + ((wave * NTW + j) * 16 + r) * 2) =
                                f2bf(acc[mt][j][i]);
                    }
        }
    }
    if (WM > 1 || sizeof(T) == 2) __syncthreads();
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 5] = __builtin_amdgcn_s_memtime();
    if (wave_active && want_stats && mhalf == 0 && q == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (nt0 + j >= NT) continue;
            float t0 = s0[j], t1 = s1[j], t2 = ss[j];
            if (WM > 1) {
                const float* d = stat_lds + ((wave * NTW + j) * 16 + r) * 3;
                t0 += d[0]; t1 += d[1]; t2 += d[2];
            }
            const int n = (nt0 + j) * 16 + r;
            if (a.bsum) {
                if (v0) a.bsum[(size_t)b0 * a.Cout + n] = t0;
                if (v1) a.bsum[(size_t)(b0 + 1) * a.Cout + n] = t1;
            }
            if (a.sqpart) a.sqpart[(size_t)blockIdx.x * a.Cout + n] = t2;
        }
    }
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 6] = __builtin_amdgcn_s_memtime();
    if constexpr (sizeof(T) == 2) {
        const int n_wg0 = blockIdx.y * BN;
        const int ncols = min(BN, a.Cout - n_wg0);   // multiple of 16
        const int ppr = ncols / 8;                   // 16-byte pieces per row
        const int rows = v1 ? kRows : (v0 ? KA_BOARD : 0);
        if (!a.ep_y) {
            for (int i = tid; i < rows * ppr; i += kThreads) {
                const int m = i / ppr, pc = i - m * ppr;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + m * ostride + pc * 16);
                *reinterpret_cast<uint4*>(static_cast<char*>(a.out) +
                                          ((size_t)(b0 * KA_BOARD + m) * a.Cout + n_wg0 + pc * 8) * 2) = v;
            }
        } else {
            // fused  da = dh*[bn(y) > 0]  and the BatchNorm-backward partial sums of da (this thread always
            // handles the same 8 channels: kThreads % ppr == 0)
            const int pc = tid % ppr, n8 = n_wg0 + pc * 8;
            float esc[8], esh[8], emu[8], eis[8], s1[8], s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                esc[e] = a.ep_scale[n8 + e]; esh[e] = a.ep_shift[n8 + e];
                emu[e] = a.ep_mean[n8 + e]; eis[e] = a.ep_invstd[n8 + e];
                s1[e] = 0.f; s2[e] = 0.f;
            }
            // all y pieces of this thread are requested up front (one HBM round trip instead of one per row)
            // (rows per thread = 162 / (kThreads/ppr) <= kMaxIt because ppr <= 32)
            constexpr int kMaxIt = (kRows + (kThreads / 32) - 1) / (kThreads / 32);
            const int mstep = kThreads / ppr, m_first = tid / ppr;
            bf16x8 yv[kMaxIt];
#pragma unroll
            for (int it = 0; it < kMaxIt; ++it) {
                const int m = m_first + it * mstep;
                yv[it] = (m < rows) ? *reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.ep_y) +
                                          ((size_t)(b0 * KA_BOARD + m) * a.Cout + n8) * 2) : bf16x8{};
            }
#pragma unroll
            for (int it = 0; it < kMaxIt; ++it) {
                const int m = m_first + it * mstep;
                if (m >= rows) continue;
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(smem + m * ostride + pc * 16);
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = (float)yv[it][e];
                    const float d = (y * esc[e] + esh[e] > 0.f) ? (float)dv[e] : 0.f;
                    s1[e] += d; s2[e] += d * ((y - emu[e]) * eis[e]);
                    ov[e] = (__bf16)d;
                }
                *reinterpret_cast<bf16x8*>(static_cast<char*>(a.out) + ((size_t)(b0 * KA_BOARD + m) * a.Cout + n8) * 2) = ov;
            }
            // combine the kThreads/ppr row-slices of every channel octet (region after the output tile + stats)
            float* red = stat_lds + WM * BN * 3;                      // [slice][ppr*8][2]
            const int slice = tid / ppr;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((slice * ppr + pc) * 8 + e) * 2] = s1[e];
                red[((slice * ppr + pc) * 8 + e) * 2 + 1] = s2[e];
            }
            __syncthreads();
            if (tid < ncols) {
                float t1 = 0.f, t2 = 0.f;
                for (int sl = 0; sl < kThreads / ppr; ++sl) { t1 += red[(sl * ncols + tid) * 2]; t2 += red[(sl * ncols + tid) * 2 + 1]; }
                a.ep_s1[(size_t)blockIdx.x * a.Cout + n_wg0 + tid] = t1;
                a.ep_s2[(size_t)blockIdx.x * a.Cout + n_wg0 + tid] = t2;
            }
        }
    } else {
        if (wave_active) {
            float* out = static_cast<float*>(a.out);
#pragma unroll
            for (int mt = 0; mt < kMTW; ++mt)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (nt0 + j >= NT) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt_base + mt) * 16 + q * 4 + i;
                        const bool ok = (m < KA_BOARD) ? v0 : (m < kRows && v1);
                        if (ok) out[(size_t)(b0 * KA_BOARD + m) * a.Cout + (nt0 + j) * 16 + r] = acc[mt][j][i];
                    }
                }
        }
    }
    if (a.stamps && tid == 0) a.stamps[wg_lin * 8 + 7] = __builtin_amdgcn_s_memtime();
}

// Pack (Co,Ci,3,3) fp32 weights into MFMA B-fragment order.
//   mode 0 (forward): out-channel n = co, in-channel c = ci (zero-padded to Ci_pad), tap = ky*3+kx
//   mode 1 (dgrad)  : out-channel n = ci, in-channel c = co, tap flipped (8 - tap)
// dst[((tap*KSG + ks)*NT + nt)*64 + lane] (16 bytes) = W'[n = nt*16 + (lane&15)][c = ks*CPK + (lane>>4)*P16 + e][tap]
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, void* __restrict__ dst, int Co, int Ci,
                                    int Nout, int Kin, int mode) {
    typedef Elem<T> E;
    constexpr int P16 = E::kPer16, CPK = 4 * P16;
    const int KSG = Kin / CPK, NT = Nout / 16;
    const size_t total = (size_t)9 * KSG * NT * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = i & 63;
        size_t t = i >> 6;
        const int nt = t % NT; t /= NT;
        const int ks = t % KSG; const int tap = t / KSG;
        const int n = nt * 16 + (lane & 15);
        float f[P16];
#pragma unroll
        for (int e = 0; e < P16; ++e) {
            const int c = ks * CPK + (lane >> 4) * P16 + e;
            float v = 0.f;
            if (mode == 0) { if (n < Co && c < Ci) v = w[((size_t)n * Ci + c) * 9 + tap]; }
            else           { if (c < Co && n < Ci) v = w[((size_t)c * Ci + n) * 9 + (8 - tap)]; }
            f[e] = v;
        }
        reinterpret_cast<typename E::vec16*>(dst)[i] = E::pack(f);
    }
}

template <typename T, int NTW, int WM>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    typedef Elem<T> E;
    const int BN = 64 * NTW;
    const size_t lds_in = (size_t)kLdsSquares * (a.KC * E::kSize + 32);
    const size_t lds_out = ((E::kSize == 2) ? (size_t)kMT * 16 * (BN * 2 + 16) : 0) + (size_t)WM * BN * 3 * sizeof(float) +
                           (a.ep_y ? (size_t)256 * WM * 16 * sizeof(float) : 0);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    KA_REQUIRE(lds <= 160 * 1024, "conv3x3: LDS tile %zu B exceeds 160 KiB (KC=%d)", lds, a.KC);
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, NTW, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            ka_set_error("conv3x3: hipFuncSetAttribute failed");
            return KA_ERR_HIP;
        }
        attr_done = true;
    }
    dim3 grid((a.B + kNB - 1) / kNB, (a.Cout + BN - 1) / BN);
    hipLaunchKernelGGL((conv3x3_kernel<T, NTW, WM>), grid, dim3(256 * WM), lds, st, a);
    return ka_check_launch("conv3x3");
}

template <typename T>
int conv_dispatch(ConvArgs a, hipStream_t st) {
    typedef Elem<T> E;
    constexpr int CPK = 4 * E::kPer16;
    KA_REQUIRE(a.B > 0 && a.Cin % CPK == 0 && a.Cout % 16 == 0,
               "conv3x3: need Cin %% %d == 0 and Cout %% 16 == 0 (got Cin=%d Cout=%d)", CPK, a.Cin, a.Cout);
    // largest LDS chunk that fits: 2 boards x 121 squares x (KC*size+16) <= 160 KiB
    int kc = a.Cin;
    while ((size_t)kLdsSquares * (kc * E::kSize + 32) > 150 * 1024) {
        KA_REQUIRE(kc % 2 == 0 && (kc / 2) % CPK == 0, "conv3x3: cannot chunk Cin=%d", a.Cin);
        kc /= 2;
    }
    if (E::kSize == 2 && kc > 128 && a.Cin % 128 == 0) kc = 128;   // measured: 2 chunks of 128 beat one of 256
    int ntw = a.Cout > 128 ? 4 : (a.Cout > 64 ? 2 : 1);
    int wm = 2;
    a.tune_prio = 1;          // measured -3 %: static priority for the second wave of every SIMD
    // tuning overrides (experiments only): channels per LDS chunk, n-tiles per wave, waves along M
    if (const char* e = getenv("KA_CONV_KC")) { const int v = atoi(e); if (v > 0 && a.Cin % v == 0 && v % CPK == 0 && v <= kc) kc = v; }
    if (const char* e = getenv("KA_CONV_NTW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) ntw = v; }
    if (const char* e = getenv("KA_CONV_WM")) { const int v = atoi(e); if (v == 1 || v == 2) wm = v; }
    if (const char* e = getenv("KA_CONV_STAGGER")) a.tune_stagger = atoi(e);
    if (const char* e = getenv("KA_CONV_PRIO")) a.tune_prio = atoi(e);
    KA_REQUIRE((256 * wm) % (kc * E::kSize / 16) == 0, "conv3x3: chunk of %d channels does not tile the workgroup", kc);
    a.KC = kc;
    if (wm == 2) {
        if (ntw == 4) return launch_conv<T, 4, 2>(a, st);
        if (ntw == 2) return launch_conv<T, 2, 2>(a, st);
        return launch_conv<T, 1, 2>(a, st);
    }
    if (ntw == 4) return launch_conv<T, 4, 1>(a, st);
    if (ntw == 2) return launch_conv<T, 2, 1>(a, st);
    return launch_conv<T, 1, 1>(a, st);
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" int ka_conv3x3_fwd(const void* in, const void* wpack, void* out, const float* in_scale,
                              const float* in_shift, const float* in_bias, int relu, float* bsum, float* sqpart,
                              int B, int Cin, int Cout, int dtype, void* stream) {
    ConvArgs a{in, wpack, out, in_scale, in_shift, in_bias, bsum, sqpart, B, Cin, Cout, 0, relu,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, g_stamps};
    KA_REQUIRE(in && wpack && out, "conv3x3: null tensor");
    KA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3: scale/shift must come together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KA_DTYPE_BF16) return conv_dispatch<bf16_t>(a, st);
    if (dtype == KA_DTYPE_F32) return conv_dispatch<float>(a, st);
    ka_set_error("conv3x3: unknown dtype %d", dtype);
    return KA_ERR_ARG;
}

// Data-gradient convolution with the surrounding BatchNorm-backward passes fused in (bf16 only):
//   input  : dy = in*k[0:C] + k[C:2C] + in2*k[2C:3C]   (ka_bn_bwd_apply on the fly; written to dy_out when non-NULL)
//   output : out = conv(dy) [* [ep_scale*ep_y + ep_shift > 0]]  and, with ep_y, the per-workgroup partial sums
//            ep_s1/ep_s2 [ka_conv3x3_sqpart_rows(B)][Cout] of ka_relu_bn_bwd_reduce; bsum = per-board sums of conv(dy)
extern "C" int ka_conv3x3_dgrad_fused(const void* in, const void* in2, const float* k, void* dy_out, const void* wpack,
                                      void* out, float* bsum, const void* ep_y, const float* ep_scale,
                                      const float* ep_shift, const float* ep_mean, const float* ep_invstd, float* ep_s1,
                                      float* ep_s2, int B, int Cin, int Cout, int dtype, void* stream) {
    KA_REQUIRE(in && in2 && k && wpack && out, "conv3x3_dgrad_fused: null tensor");
    KA_REQUIRE(dtype == KA_DTYPE_BF16, "conv3x3_dgrad_fused: bf16 only");
    KA_REQUIRE(!ep_y || (ep_scale && ep_shift && ep_mean && ep_invstd && ep_s1 && ep_s2), "conv3x3_dgrad_fused: epilogue tensors");
    ConvArgs a{in, wpack, out, k, k + Cin, nullptr, bsum, nullptr, B, Cin, Cout, 0, 0,
               in2, k + 2 * Cin, dy_out, ep_y, ep_scale, ep_shift, ep_mean, ep_invstd, ep_s1, ep_s2, 0, 0, g_stamps};
    return conv_dispatch<bf16_t>(a, static_cast<hipStream_t>(stream));
}

// diagnostic: stamps != null makes every conv3x3 workgroup record 4 s_memtime values (100 MHz ticks are NOT used:
// s_memtime counts shader clocks); pass null to switch off
extern "C" int ka_debug_conv_stamps(unsigned long long* stamps) { g_stamps = stamps; return KA_OK; }

extern "C" int ka_conv3x3_sqpart_rows(int B) { return (B + kNB - 1) / kNB; }

extern "C" int ka_pack_conv3x3(const float* w, void* dst, int Co, int Ci, int Nout, int Kin, int mode, int dtype,
                               void* stream) {
    KA_REQUIRE(w && dst && (mode == 0 || mode == 1), "pack_conv3x3: bad arguments");
    KA_REQUIRE(Nout % 16 == 0, "pack_conv3x3: Nout %% 16 != 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cpk = dtype == KA_DTYPE_BF16 ? 32 : 16;
    KA_REQUIRE(Kin % cpk == 0, "pack_conv3x3: Kin %% %d != 0", cpk);
    const size_t total = (size_t)9 * (Kin / cpk) * (Nout / 16) * 64;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dtype == KA_DTYPE_BF16)
        hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w, dst, Co, Ci, Nout, Kin, mode);
    else if (dtype == KA_DTYPE_F32)
        hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(blocks), dim3(256), 0, st, w, dst, Co, Ci, Nout, Kin, mode);
    else { ka_set_error("pack_conv3x3: unknown dtype %d", dtype); return KA_ERR_ARG; }
    return ka_check_launch("pack_conv3x3");
}
