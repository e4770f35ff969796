// 3x3 "same" convolution over 9x9 boards as an implicit GEMM on the CDNA4 matrix cores.
//
//   out[b,p,n] = sum_{tap,c} X'[b, p+tap, c] * W[n,c,tap]      X' = optional fused transform of the input
//
// Layout: activations are NHWC (B,81,C), channel-contiguous, so every HBM access is a
// coalesced run along the channel axis.  One 256-thread workgroup (4 waves, one per SIMD) owns
// ONE whole board and a slab of output channels, and two such workgroups share a CU: they are
// independent, so the staging and epilogue of one run under the MFMA loop of the other (the
// 512-thread form with two boards per workgroup is kept behind KA_CONV_WM=2).  The board is
// staged ONCE into LDS as a zero-haloed 17-wide image ([padded square][channel]), after which
// each of the 9 taps is just a constant LDS offset ((dy*17+dx)*row_stride) on the MFMA operand
// read -- the input is read from HBM exactly once per output-channel slab.  Weights are
// pre-packed (pack_conv3x3_kernel) in MFMA fragment order so a wave streams them from L2 with
// perfectly coalesced 1 KiB loads straight into registers; no weight goes through LDS.
//
// Wave tile: wave w owns the board's 81 squares, padded to 6 row tiles of 16, and the w-th
// quarter of the channel slab.  The weights are the MFMA "A" operand and the activations the
// "B" operand, i.e. the kernel computes out^T: an accumulator lane then holds 4 CONSECUTIVE
// channel slots of one square, and the pack interleaves neighbouring 16-channel tiles so that
// a lane owns runs of 8 consecutive channels.  The epilogue therefore works entirely in
// registers: the per-board channel sums (= SE squeeze, and summed over boards the BN mean) and
// sums of squares are DPP reductions over the 16 square lanes of a wave that owns the whole
// board, and the output tile is stored straight from the accumulators in 16-byte pieces -- no
// LDS transpose and no barrier after the main loop.
// (v_mfma_f32_16x16x32_bf16 / 4x v_mfma_f32_16x16x4_f32; the f32 path is exact f32 (parity
// mode), the bf16 path is the throughput mode.)
//
// The same kernel is the data-gradient conv: pack with flip+transpose (mode 1).
//
// Reference semantics replaced: nn.Conv2d(C, C, 3, padding=1, bias=False) at
// keisei/training/models/se_resnet.py:50,52,110 and its autograd backward.
#include <stdlib.h>
#include <utility>
#include "common.h"

namespace {

constexpr int kNB = 2;                       // boards per workgroup in the paired layout
// LDS image of the input: zero-haloed boards, [square][channel].  A padded board row is 17 squares wide (9 + halo,
// widened so that stepping to the next board row advances the square index by 8 more than a neighbour step) and the
// second board starts 193 squares after the first; with a row stride of 32 bytes more than a multiple of 256 a
// 16-lane MFMA fragment read (16 consecutive GEMM rows x 16 bytes) then lands on 16 different 16-byte bank slots for
// every tap -- conflict-free ds_read_b128, where the natural 11-wide image costs a 2-way conflict on every read.
constexpr int kPW = 17;                      // squares per padded board row
constexpr int kBoardStride = 193;            // squares between the two boards
constexpr int kLdsSquares = kBoardStride + 11 * kPW;   // 380 (two boards)
// single-board image: the squares any tap of any board square can address are 0 .. 10 * kPW + 10
constexpr int kImgSquares1 = 10 * kPW + 11;            // 181
constexpr int kZeroSquares = 2 * (kPW + 1) + 1;        // all-zero squares behind the image: the taps of its centre square (37)
__device__ __forceinline__ int lds_square(int b, int p) { return b * kBoardStride + (p / 9 + 1) * kPW + (p % 9) + 1; }
constexpr int kMTW = 6;                      // row tiles per wave: one board, 81 squares padded to 96 rows

// output-channel slot permutation of the weight pack: MFMA row s of tile nt multiplies output channel chan_of(nt, s).
// The two tiles of a pair are interleaved in groups of 4 so that accumulator lane q (rows 4q..4q+3 of both tiles)
// owns the 8 consecutive channels pair*32 + 8q .. +7.  A lone last tile keeps the identity order.
__host__ __device__ __forceinline__ int chan_of(int nt, int s, int NT) {
    return ((nt | 1) < NT) ? (nt >> 1) * 32 + (s >> 2) * 8 + (nt & 1) * 4 + (s & 3) : nt * 16 + s;
}

// sum over the 16 lanes of a DPP row (lanes that share lane>>4); every lane ends up with the total
__device__ __forceinline__ float row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm 1,0,3,2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm 2,3,0,1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

struct ConvArgs {
    const void* in;
    const void* wpack;
    void* out;
    const float* in_scale;   // [Cin] or null: x' = x*scale + shift
    const float* in_shift;
    const float* in_bias;    // [B,Cin] or null: per-board bias added after the ReLU.  WITH in2 (two-tensor form) it is instead
                             // gate_add = [gate | add], [2][B][Cin]: `in` is du and the gradient the transform applies to is
                             // dz = in*gate[b,c] + add[b,c] (ka_conv3x3_dgrad_fused_gated), x' = dz*in_scale + in_shift + in2*in_k3 with dz
                             // never rounded to bf16 nor written; conv3x3_pc2_kernel<GATED> (which also takes in_shift / in_k3 as
                             // in_scale + Cin / 2 Cin) and conv3x3_corner_kernel only
    float* bsum;             // [B,Cout] or null
    float* sqpart;           // [B, Cout] or null: per-board sums of squares
    int B, Cin, Cout, KC, relu;
    // fused BatchNorm-backward input (data-gradient convs): x' = in*in_scale + in_shift + in2*in_k3, optionally
    // written back to in_out (the materialised dy the weight-gradient kernel reads)
    const void* in2;
    const float* in_k3;
    void* in_out;
    // fused ReLU+BatchNorm-backward epilogue (bf16): out = acc * [ep_scale*ep_y + ep_shift > 0]; per-board partial
    // sums ep_s1[b][n] = sum out, ep_s2[b][n] = sum out*(ep_y-ep_mean)*ep_invstd
    const void* ep_y;
    const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_invstd;
    float* ep_s1; float* ep_s2;
    int tune_stagger, tune_prio;  // experiments: s_sleep count / static priority of the second wave of every SIMD
    unsigned long long* stamps;   // diagnostic only: [workgroup][8] s_memtime at phase boundaries (null in production)
    int mt5;                      // five row tiles (squares 0..79) in the tower kernels, square 80 by conv3x3_corner_kernel
};

std::atomic<unsigned long long*>& g_stamps = ka_debug_stamps();   // diagnostic only (ka_debug_conv_stamps); null in production

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

// ---- epilogue, all in registers: lane (r, q) holds square mt*16+r of board bb for every row tile mt, and for
// tile j the 4 consecutive channels cb[j] .. cb[j]+3 (tiles 2k and 2k+1 together: 8 consecutive channels)
template <typename T, int NTW, int MT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NTW], int bb, int nt0, int NT, int r, int q) {
    int cb[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) cb[j] = chan_of(min(nt0 + j, NT - 1), 4 * q, NT);
    // masked epilogue: the y pieces of the first tile pair are requested BEFORE the statistics arithmetic, which
    // then covers their HBM latency (requesting both pairs up front spills and is slower)
    constexpr int TP0 = NTW >= 2 ? 2 : 1;
    typedef __attribute__((ext_vector_type(4 * TP0))) __bf16 bvec0;
    bvec0 yv0[MT];
    if constexpr (sizeof(T) == 2) {
        if (a.ep_y) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = mt * 16 + r;
                yv0[mt] = bvec0{};
                if (p < KA_BOARD)
                    yv0[mt] = *reinterpret_cast<const bvec0*>(static_cast<const char*>(a.ep_y) +
                                                              ((size_t)(bb * KA_BOARD + p) * a.Cout + cb[0]) * 2);
            }
        }
    }
    if (a.bsum || a.sqpart) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            float s0[4], ss[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = 0.f; ss[i] = 0.f; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool in = mt * 16 + r < KA_BOARD;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = in ? acc[mt][j][i] : 0.f;
                    s0[i] += v; ss[i] += v * v;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = row_sum16(s0[i]); ss[i] = row_sum16(ss[i]); }
            if (r == 0 && nt0 + j < NT) {
                if (a.bsum) *reinterpret_cast<f32x4*>(a.bsum + (size_t)bb * a.Cout + cb[j]) = f32x4{s0[0], s0[1], s0[2], s0[3]};
                if (a.sqpart) *reinterpret_cast<f32x4*>(a.sqpart + (size_t)bb * a.Cout + cb[j]) = f32x4{ss[0], ss[1], ss[2], ss[3]};
            }
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (!a.ep_y) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = mt * 16 + r;
                if (p >= KA_BOARD) continue;
                char* orow = static_cast<char*>(a.out) + (size_t)(bb * KA_BOARD + p) * a.Cout * 2;
                if constexpr (NTW >= 2) {
#pragma unroll
                    for (int j = 0; j < NTW; j += 2) {
                        bf16x8 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { o[i] = (__bf16)acc[mt][j][i]; o[4 + i] = (__bf16)acc[mt][j + 1][i]; }
                        if (nt0 + j < NT) *reinterpret_cast<bf16x8*>(orow + cb[j] * 2) = o;
                    }
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)acc[mt][0][i];
                    *reinterpret_cast<bf16x4*>(orow + cb[0] * 2) = o;
                }
            }
        } else {
            // fused  da = dh*[bn(y) > 0]  and the BatchNorm-backward partial sums of da, one PAIR of tiles (8
            // consecutive channels, 16-byte accesses) at a time; a lone tile (NTW == 1) uses 8-byte accesses
            constexpr int TP = NTW >= 2 ? 2 : 1, NE = 4 * TP;
            typedef __attribute__((ext_vector_type(NE))) __bf16 bvec;
#pragma unroll
            for (int j = 0; j < NTW; j += TP) {
                if (nt0 + j >= NT) continue;
                bvec yv[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = mt * 16 + r;
                    if (j == 0) { yv[mt] = yv0[mt]; continue; }
                    yv[mt] = bvec{};
                    if (p < KA_BOARD)
                        yv[mt] = *reinterpret_cast<const bvec*>(static_cast<const char*>(a.ep_y) +
                                                                ((size_t)(bb * KA_BOARD + p) * a.Cout + cb[j]) * 2);
                }
                float esc[NE], esh[NE], emu[NE], eis[NE], t1[NE], t2[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    esc[e] = a.ep_scale[cb[j] + e]; esh[e] = a.ep_shift[cb[j] + e];
                    emu[e] = a.ep_mean[cb[j] + e]; eis[e] = a.ep_invstd[cb[j] + e];
                    t1[e] = 0.f; t2[e] = 0.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = mt * 16 + r;
                    const bool in = p < KA_BOARD;
                    bvec o;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const float y = (float)yv[mt][e];
                        const __bf16 db = (__bf16)acc[mt][j + (e >> 2)][e & 3];
                        const float d = (in && y * esc[e] + esh[e] > 0.f) ? (float)db : 0.f;
                        t1[e] += d; t2[e] += d * ((y - emu[e]) * eis[e]);
                        o[e] = (__bf16)d;
                    }
                    if (in) *reinterpret_cast<bvec*>(static_cast<char*>(a.out) + ((size_t)(bb * KA_BOARD + p) * a.Cout + cb[j]) * 2) = o;
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) { t1[e] = row_sum16(t1[e]); t2[e] = row_sum16(t2[e]); }
                if (r == 0) {
#pragma unroll
                    for (int e = 0; e < NE; e += 4) {
                        *reinterpret_cast<f32x4*>(a.ep_s1 + (size_t)bb * a.Cout + cb[j] + e) = f32x4{t1[e], t1[e + 1], t1[e + 2], t1[e + 3]};
                        *reinterpret_cast<f32x4*>(a.ep_s2 + (size_t)bb * a.Cout + cb[j] + e) = f32x4{t2[e], t2[e + 1], t2[e + 2], t2[e + 3]};
                    }
                }
            }
        }
    } else {
        float* out = static_cast<float*>(a.out);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = mt * 16 + r;
            if (p >= KA_BOARD) continue;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                if (nt0 + j < NT) *reinterpret_cast<f32x4*>(out + (size_t)(bb * KA_BOARD + p) * a.Cout + cb[j]) = acc[mt][j];
        }
    }
}

// NTW = 16-channel tiles per wave (the workgroup's slab is 4*NTW tiles wide).  WM = boards (= groups of 4 waves) per
// workgroup: 2 -> 512 threads, the second wave of every SIMD owns the second board and its MFMAs fill the first one's
// LDS/L2 stalls; 1 -> 256 threads and two INDEPENDENT workgroups per CU (registers capped at 256 by the launch
// bounds), so that the staging and epilogue of one overlap the MFMA loop of the other.
template <typename T, int NTW, int WM, int MT>
__global__ __launch_bounds__(256 * WM, 2) void conv3x3_kernel(ConvArgs a) {
    constexpr int kThreads = 256 * WM;
    constexpr int kRows = WM * KA_BOARD;         // staged squares
    constexpr int kImgSquares = WM == 2 ? kLdsSquares : kImgSquares1;
    typedef Elem<T> E;
    typedef typename E::vec16 vec16;
    constexpr int ESZ = E::kSize, P16 = E::kPer16, CPK = 4 * P16;   // channels per k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, mhalf = tid >> 8;
    const int r = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * WM;
    const int NT = a.Cout >> 4;
    const int nt0 = blockIdx.y * (4 * NTW) + wave * NTW;
    const int stride = a.KC * ESZ + 32;          // bytes per padded square (+32: bank spread, see kPW)
    const int cpr = a.KC * ESZ / 16;             // 16-byte pieces per square
    const int KSG = a.Cin / CPK;                 // k-steps over all input channels
    const int KS = a.KC / CPK;                   // k-steps per LDS chunk

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int p = mt * 16 + r;                     // square of board `mhalf`
        // dummy rows (81 -> 96 padding; never stored) read an all-zero region behind the image: every tap offset of its
        // centre square stays inside it.  Zero operands cost the matrix pipe less power than duplicates of square 0,
        // and under this kernel the chip's clock is what the power budget leaves.
        rowoff[mt] = (p >= KA_BOARD ? (kImgSquares + kPW + 1) : lds_square(mhalf, p)) * stride + q * 16;
    }
    for (int i = tid; i < kZeroSquares * cpr; i += kThreads)
        *reinterpret_cast<uint4*>(smem + (kImgSquares + i / cpr) * stride + (i % cpr) * 16) = uint4{0, 0, 0, 0};

    // zero the halo once (staging only ever writes interior squares)
    for (int i = tid; i < kImgSquares * cpr; i += kThreads) {
        const int idx = i / cpr, j = i - idx * cpr;
        const int pp = idx >= kBoardStride ? idx - kBoardStride : idx, yy = pp / kPW, xx = pp - yy * kPW;
        const bool interior = yy >= 1 && yy <= 9 && xx >= 1 && xx <= 9;
        if (!interior) *reinterpret_cast<uint4*>(smem + idx * stride + j * 16) = uint4{0, 0, 0, 0};
    }

    const int sj = tid % cpr, spos0 = tid / cpr, sstep = kThreads / cpr;   // staging role of this thread
    const char* wbase = static_cast<const char*>(a.wpack) + ((size_t)nt0 * 64 + lane) * 16;
    const size_t tap_stride = (size_t)KSG * NT * 1024, ks_stride = (size_t)NT * 1024;
    const bool wave_active = nt0 < NT;

    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (a.stamps && tid == 0) {
        a.stamps[wg_lin * 8 + 0] = __builtin_amdgcn_s_memtime();
        a.stamps[wg_lin * 8 + 3] = __builtin_amdgcn_s_memrealtime();      // 100 MHz: with slot 4, the clock the kernel held
    }
    const int nchunks = a.Cin / a.KC;
    for (int kc = 0; kc < nchunks; ++kc) {
        // (the barrier that frees the image for the next chunk sits INSIDE the staging code, between the global loads of the
        // first batch and their LDS writes: a wave that arrives early waits for the others with its loads already in flight)
        // ---- stage NB boards x KC channels into the haloed LDS tile (fused input transform)
        {
            const int c0 = kc * a.KC + sj * P16;
            float sc[P16], sh[P16];
            const bool has_aff = a.in_scale != nullptr;
            if (has_aff) {
#pragma unroll
                for (int e = 0; e < P16; ++e) { sc[e] = a.in_scale[c0 + e]; sh[e] = a.in_shift[c0 + e]; }
            }
            float gbv[WM][P16];                 // per-board bias of this thread's channel piece, fetched once
            if (a.in_bias) {
#pragma unroll
                for (int b = 0; b < WM; ++b)
#pragma unroll
                    for (int e = 0; e < P16; ++e)
                        gbv[b][e] = (b0 + b < a.B) ? a.in_bias[(size_t)(b0 + b) * a.Cin + c0 + e] : 0.f;
            }
            if (a.in2) {
                // two-tensor transform (BatchNorm backward applied on the fly), 3 + 3 loads in flight
                float k3[P16];
#pragma unroll
                for (int e = 0; e < P16; ++e) k3[e] = a.in_k3[c0 + e];
                constexpr int kU2 = 6;
                for (int pos0 = spos0; pos0 < kRows; pos0 += sstep * kU2) {
                    vec16 v[kU2], w[kU2];
#pragma unroll
                    for (int u = 0; u < kU2; ++u) {
                        const int pos = pos0 + u * sstep;
                        const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                        const size_t off = ((size_t)(bb * KA_BOARD + p) * a.Cin + c0) * ESZ;
                        if (pos < kRows && bb < a.B) {
                            v[u] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(static_cast<const char*>(a.in) + off));
                            w[u] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(static_cast<const char*>(a.in2) + off));
                        } else { v[u] = vec16{}; w[u] = vec16{}; }
                    }
                    if (kc > 0 && pos0 == spos0) KA_LDS_BARRIER();      // every thread runs this first batch exactly once
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
            constexpr int kUnr = 6;
            for (int pos0 = spos0; pos0 < kRows; pos0 += sstep * kUnr) {
                vec16 v[kUnr];
#pragma unroll
                for (int u = 0; u < kUnr; ++u) {
                    const int pos = pos0 + u * sstep;
                    const int b = pos / KA_BOARD, p = pos - b * KA_BOARD, bb = b0 + b;
                    if (pos < kRows && bb < a.B)
                        v[u] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(static_cast<const char*>(a.in) +
                                                               ((size_t)(bb * KA_BOARD + p) * a.Cin + c0) * ESZ));
                    else
                        v[u] = vec16{};
                }
                if (kc > 0 && pos0 == spos0) KA_LDS_BARRIER();          // every thread runs this first batch exactly once
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
                            for (int e = 0; e < P16; ++e) f[e] += (WM == 1 || b == 0) ? gbv[0][e] : gbv[WM - 1][e];
                        }
                        v[u] = E::pack(f);
                    }
                    *reinterpret_cast<vec16*>(smem + lds_square(b, p) * stride + sj * 16) = v[u];
                }
            }
                    }
        }
        // the first k-step's weight fragments of this chunk are requested BEFORE the barrier that publishes the image and
        // stay in flight across it (the barrier orders LDS traffic only): the MFMA phase no longer starts with an L2 round trip
        vec16 bpre[NTW];
        {
            const char* wp = wbase + (size_t)(kc * (a.KC / CPK)) * ks_stride;
#pragma unroll
            for (int j = 0; j < NTW; ++j) bpre[j] = *reinterpret_cast<const vec16*>(wp + (min(nt0 + j, NT - 1) - nt0) * 1024);
        }
        KA_LDS_BARRIER();
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
            auto load_a = [&](vec16 (&av)[MT], int toff) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const vec16*>(smem + rowoff[mt] + toff);
            };
            auto compute = [&](const vec16 (&bw)[NTW], const vec16 (&av)[MT], vec16 (&anext)[MT], int toff_next) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#ifndef KA_DIAG_NO_A
                    anext[mt] = *reinterpret_cast<const vec16*>(smem + rowoff[mt] + toff_next);
#else
                    anext[mt] = av[mt];
#endif
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[mt][j] = Mma<T>::run(bw[j], av[mt], acc[mt][j]);
                }
                // pin the issue order: one LDS read ahead of every group of MFMAs (the scheduler otherwise
                // sinks all 11 reads behind the MFMA block and the next step starts by waiting for them)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NTW * (sizeof(T) == 2 ? 1 : 4), 0);
                }
            };
            vec16 b0[NTW], b1[NTW], a0[MT], a1[MT];
#pragma unroll
            for (int j = 0; j < NTW; ++j) b0[j] = bpre[j];
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
    const int bb = b0 + mhalf;
    if (wave_active && bb < a.B) conv_epilogue<T, NTW, MT>(a, acc, bb, nt0, NT, r, q);
    if (a.stamps && tid == 0) {
        a.stamps[wg_lin * 8 + 7] = __builtin_amdgcn_s_memtime();
        a.stamps[wg_lin * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    }
}

// Pack (Co,Ci,3,3) fp32 weights into MFMA fragment order.
//   mode 0 (forward): out-channel n = co, in-channel c = ci (zero-padded to Ci_pad), tap = ky*3+kx
//   mode 1 (dgrad)  : out-channel n = ci, in-channel c = co, tap flipped (8 - tap)
// dst[((tap*KSG + ks)*NT + nt)*64 + lane] (16 bytes) = W'[n = chan_of(nt, lane&15)][c = ks*CPK + (lane>>4)*P16 + e][tap]
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
        const int n = chan_of(nt, lane & 15, NT);
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

// All layers of a network in one launch: table[n][8] (int64) = {src, dst, Co, Ci, Nout, Kin, mode, unused}; blockIdx.y
// picks the layer.  (80 tower convolutions x {forward, dgrad} packs after every optimiser step: as single launches
// they are CPU-launch-bound and leave the GPU idle for ~2 ms at the start of a step.)
template <typename T>
__global__ void pack_conv3x3_multi_kernel(const long long* __restrict__ table) {
    typedef Elem<T> E;
    constexpr int P16 = E::kPer16, CPK = 4 * P16;
    const long long* t = table + (size_t)blockIdx.y * 8;
    const float* __restrict__ w = reinterpret_cast<const float*>(t[0]);
    void* __restrict__ dst = reinterpret_cast<void*>(t[1]);
    const int Co = (int)t[2], Ci = (int)t[3], Nout = (int)t[4], Kin = (int)t[5], mode = (int)t[6];
    const int KSG = Kin / CPK, NT = Nout / 16;
    // one thread per (ks, nt, lane): it produces the piece of ALL nine taps, so that its weight reads are runs of nine
    // consecutive floats (forward pack: P16 such runs back to back = one contiguous stretch) instead of nine passes
    // over the weights with a stride of nine
    const size_t per_tap = (size_t)KSG * NT * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_tap; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = i & 63;
        size_t u = i >> 6;
        const int nt = u % NT;
        const int ks = u / NT;
        const int n = chan_of(nt, lane & 15, NT);
        // (all 9 * P16 reads requested together from an address that is always valid -- a channel past the tensor reads element 0
        //  and is zeroed afterwards: behind `ok ? src[..] : 0` each read sat behind a branch, one round trip after the other)
        float f[9][P16];
#pragma unroll
        for (int e = 0; e < P16; ++e) {
            const int c = ks * CPK + (lane >> 4) * P16 + e;
            const bool ok = mode == 0 ? (n < Co && c < Ci) : (c < Co && n < Ci);
            const float* src = !ok ? w : (mode == 0 ? w + ((size_t)n * Ci + c) * 9 : w + ((size_t)c * Ci + n) * 9);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) f[tap][e] = src[mode == 0 ? tap : 8 - tap];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) f[tap][e] = ok ? f[tap][e] : 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            reinterpret_cast<typename E::vec16*>(dst)[(size_t)tap * per_tap + i] = E::pack(f[tap]);
    }
}

template <typename T, int NTW, int WM>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    typedef Elem<T> E;
    const int BN = 64 * NTW;
    const size_t lds = (size_t)((WM == 2 ? kLdsSquares : kImgSquares1) + kZeroSquares) * (a.KC * E::kSize + 32);
    KA_REQUIRE(lds <= 160 * 1024, "conv3x3: LDS tile %zu B exceeds 160 KiB (KC=%d)", lds, a.KC);
    dim3 grid((a.B + WM - 1) / WM, (a.Cout + BN - 1) / BN);
    if constexpr (sizeof(T) == 2 && NTW == 4 && WM == 1) {
        if (a.mt5) {
            static std::atomic<unsigned long long> attr5{0};
            if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv3x3_kernel<T, NTW, WM, 5>), attr5, "conv3x3 (5 row tiles)")) return rc;
            hipLaunchKernelGGL((conv3x3_kernel<T, NTW, WM, 5>), grid, dim3(256 * WM), lds, st, a);
            return ka_check_launch("conv3x3 (5 row tiles)");
        }
    }
    KA_REQUIRE(!a.mt5, "conv3x3: the five-row-tile form exists for bf16, 4 tiles per wave, one board per workgroup");
    static std::atomic<unsigned long long> attr_done{0};   // per instantiation: devices already configured
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv3x3_kernel<T, NTW, WM, kMTW>), attr_done, "conv3x3")) return rc;
    hipLaunchKernelGGL((conv3x3_kernel<T, NTW, WM, kMTW>), grid, dim3(256 * WM), lds, st, a);
    return ka_check_launch("conv3x3");
}

constexpr int kStImgStride = 128 * 2 + 32;                 // bytes per square of a 128-channel LDS image (32 B more than a multiple of 256)
// ---------------------------------------------------------------- producer / consumer form (bf16, Cin = Cout = 256)
// What keeps conv3x3_kernel at ~365 us is that staging (HBM latency + input transform + LDS
// writes) and the MFMA phases of a workgroup are serial; two workgroups per CU overlap them only by chance.  Here the
// overlap is explicit: a persistent 768-thread workgroup per CU, waves 0-7 multiply (the eval tower's loop, weights
// through the register ring, continuous across units), waves 8-11 stage the NEXT unit -- one (board, 128-channel chunk)
// -- straight into the other of two zero-haloed LDS images while the current one is multiplied.  One LDS-only barrier
// per unit.  The input transforms (BatchNorm + ReLU + bias, or the two-tensor data-gradient input and its write-back)
// live in the staging waves, whose registers are otherwise idle -- the two-tensor form spilled when the MFMA waves had
// to carry it.  Same ConvArgs, weight packs and epilogue as conv3x3_kernel: bit-identical results.
constexpr int kPcImg = kImgSquares1 * kStImgStride;                       // one 128-channel image: 181 squares x 288 B
constexpr int kPcZero = 2 * kPcImg;                                       // the all-zero squares the padded rows read
constexpr int kPcLds = kPcZero + kZeroSquares * kStImgStride;
constexpr int kPcYStride = 256 * 2 + 16;                                  // the masked epilogue's y rows in LDS (16 B of padding: 2-way conflicts at most)
constexpr int kPcLdsMasked = kPcLds + KA_BOARD * kPcYStride;

// STAG: MFMA waves 4-7 (the SIMD partners of waves 0-3) run half a unit behind waves 0-3, two barriers per unit: the epilogue and
// the unit-start latencies of one wave then sit beside the matrix work of its partner instead of beside the partner's own
// (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  An image is read for three half-unit slots, so the staging waves load
// the next unit's pieces in one slot and write them to LDS in the following one, the only slot in which that image is free.
template <bool TWO, bool MASKED, int NPW = 4, int MT = 6, bool STAG = false>     // TWO: the two-tensor data-gradient input; MASKED: its ReLU + BatchNorm-backward epilogue; NPW staging waves; MT row tiles
__global__ __launch_bounds__(512 + 64 * NPW) void conv3x3_pc_kernel(ConvArgs a) {
    static_assert(!(STAG && MASKED), "the staggered schedule is built for the register-only epilogue");
    constexpr int NT_ = 512 + 64 * NPW, NP = 64 * NPW;
    constexpr int KP = (KA_BOARD * 16 + NP - 1) / NP, KY = (KA_BOARD * 32 + NP - 1) / NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x;
    if ((int)blockIdx.x >= a.B) return;
    for (int i = tid; i < kPcLds / 16; i += NT_) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
    const int nboards = (a.B - (int)blockIdx.x + nwg - 1) / nwg, nunits = 2 * nboards;
    if (!MASKED) a.ep_y = nullptr;                           // (compile-time: the other epilogue is not compiled in)
    __syncthreads();

    if (wave >= 8) {
        // ---------------- staging waves: piece i = pt + NP k of a unit = row i / 16, 16-byte piece i % 16 = pt % 16
        const int pt = tid - 512, pc = pt & 15;
        const bool has_aff = a.in_scale != nullptr;
        // (the pieces of a unit live in registers across a barrier only in the staggered schedule: `stage` keeps its own,
        //  local to one call -- as outer arrays they are kept in scratch across the unit loop's back-edge)
        auto stage_load = [&](int u, bf16x8 (&pv)[KP], bf16x8 (&pw)[TWO ? KP : 1]) {
            const int bb = (int)blockIdx.x + (u >> 1) * nwg, kc = u & 1, ch0 = kc * 128 + pc * 8;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = pt + NP * k;
                pv[k] = bf16x8{};
                if (TWO) pw[TWO ? k : 0] = bf16x8{};
                if (i < KA_BOARD * 16) {
                    const size_t off = (((size_t)bb * KA_BOARD + (i >> 4)) * 256 + ch0) * 2;
                    pv[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in) + off));
                    if (TWO) pw[TWO ? k : 0] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in2) + off));
                }
            }
        };
        auto stage_write = [&](int u, bf16x8 (&pv)[KP], bf16x8 (&pw)[TWO ? KP : 1]) {
            const int bb = (int)blockIdx.x + (u >> 1) * nwg, kc = u & 1, ch0 = kc * 128 + pc * 8;
            char* img = smem + (u & 1) * kPcImg;
            float sc[8], sh[8], k3[8], pb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = has_aff ? a.in_scale[ch0 + e] : 1.f; sh[e] = has_aff ? a.in_shift[ch0 + e] : 0.f;
                k3[e] = TWO ? a.in_k3[ch0 + e] : 0.f;
                pb[e] = (!TWO && a.in_bias) ? a.in_bias[(size_t)bb * 256 + ch0 + e] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = pt + NP * k;
                if (i >= KA_BOARD * 16) continue;
                bf16x8 v = pv[k];
                if (TWO) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaf((float)pw[TWO ? k : 0][e], k3[e], fmaf((float)v[e], sc[e], sh[e]));
                    if (a.in_out) *reinterpret_cast<bf16x8*>(static_cast<char*>(a.in_out) + (((size_t)bb * KA_BOARD + (i >> 4)) * 256 + ch0) * 2) = v;
                } else if (has_aff || a.relu || a.in_bias) {
                    // two channels per instruction (v_pk_fma_f32 / v_pk_add_f32): this arithmetic shares the SIMDs with the MFMA waves
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 f = {(float)v[e], (float)v[e + 1]};
                        if (has_aff) f = __builtin_elementwise_fma(f, f32x2{sc[e], sc[e + 1]}, f32x2{sh[e], sh[e + 1]});
                        if (a.relu) f = __builtin_elementwise_max(f, f32x2{0.f, 0.f});
                        if (a.in_bias) f += f32x2{pb[e], pb[e + 1]};
                        v[e] = (__bf16)f[0]; v[e + 1] = (__bf16)f[1];
                    }
                }
                *reinterpret_cast<bf16x8*>(img + lds_square(0, i >> 4) * kStImgStride + pc * 16) = v;
            }
        };
        auto stage = [&](int u) {
            const int bb = (int)blockIdx.x + (u >> 1) * nwg, kc = u & 1, ch0 = kc * 128 + pc * 8;
            char* img = smem + (u & 1) * kPcImg;
            bf16x8 pv[KP], pw[TWO ? KP : 1];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = pt + NP * k;
                pv[k] = bf16x8{};
                if (TWO) pw[TWO ? k : 0] = bf16x8{};
                if (i < KA_BOARD * 16) {
                    const size_t off = (((size_t)bb * KA_BOARD + (i >> 4)) * 256 + ch0) * 2;
                    pv[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in) + off));
                    if (TWO) pw[TWO ? k : 0] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in2) + off));
                }
            }
            float sc[8], sh[8], k3[8], pb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = has_aff ? a.in_scale[ch0 + e] : 1.f; sh[e] = has_aff ? a.in_shift[ch0 + e] : 0.f;
                k3[e] = TWO ? a.in_k3[ch0 + e] : 0.f;
                pb[e] = (!TWO && a.in_bias) ? a.in_bias[(size_t)bb * 256 + ch0 + e] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = pt + NP * k;
                if (i >= KA_BOARD * 16) continue;
                bf16x8 v = pv[k];
                if (TWO) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaf((float)pw[TWO ? k : 0][e], k3[e], fmaf((float)v[e], sc[e], sh[e]));
                    if (a.in_out) *reinterpret_cast<bf16x8*>(static_cast<char*>(a.in_out) + (((size_t)bb * KA_BOARD + (i >> 4)) * 256 + ch0) * 2) = v;
                } else if (has_aff || a.relu || a.in_bias) {
                    // two channels per instruction (v_pk_fma_f32 / v_pk_add_f32): this arithmetic shares the SIMDs with the MFMA waves
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 f = {(float)v[e], (float)v[e + 1]};
                        if (has_aff) f = __builtin_elementwise_fma(f, f32x2{sc[e], sc[e + 1]}, f32x2{sh[e], sh[e + 1]});
                        if (a.relu) f = __builtin_elementwise_max(f, f32x2{0.f, 0.f});
                        if (a.in_bias) f += f32x2{pb[e], pb[e + 1]};
                        v[e] = (__bf16)f[0]; v[e + 1] = (__bf16)f[1];
                    }
                }
                *reinterpret_cast<bf16x8*>(img + lds_square(0, i >> 4) * kStImgStride + pc * 16) = v;
            }
        };
        // masked form: the y rows the epilogue of board b compares against are brought into LDS during the board's second
        // unit, and the MFMA waves run that epilogue right AFTER the barrier that ends it (no HBM latency, few registers)
        auto load_y = [&](int bb) {
            bf16x8 yv[KY];
#pragma unroll
            for (int k = 0; k < KY; ++k) {
                const int i = pt + NP * k;                   // 81 rows x 32 pieces
                yv[k] = bf16x8{};
                if (i < KA_BOARD * 32)
                    yv[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.ep_y) + ((size_t)bb * KA_BOARD * 32 + i) * 16));
            }
#pragma unroll
            for (int k = 0; k < KY; ++k) {
                const int i = pt + NP * k;
                if (i < KA_BOARD * 32) *reinterpret_cast<bf16x8*>(smem + kPcLds + (i >> 5) * kPcYStride + (i & 31) * 16) = yv[k];
            }
        };
        stage(0);
        KA_LDS_BARRIER();
        if constexpr (STAG) {
            // slot 2v-2: the loads of unit v; slot 2v-1: its LDS writes (waves 4-7 read the image's previous unit until then)
            for (int v = 1; v < nunits; ++v) {
                bf16x8 pv[KP], pw[TWO ? KP : 1];
                stage_load(v, pv, pw);
                KA_LDS_BARRIER();
                stage_write(v, pv, pw);
                KA_LDS_BARRIER();
            }
            KA_LDS_BARRIER();
            KA_LDS_BARRIER();
            return;
        }
        for (int u = 0; u < nunits; ++u) {
            if (u + 1 < nunits && !(a.tune_stagger & 2)) stage(u + 1);
            if (MASKED && (u & 1)) load_y((int)blockIdx.x + (u >> 1) * nwg);
            KA_LDS_BARRIER();
        }
        return;
    }

    // ---------------- MFMA waves: 6 row tiles x 2 channel tiles each, the weight ring runs on across the units
    if (a.tune_prio == 1) __builtin_amdgcn_s_setprio(1);
    if (a.tune_prio >= 2) __builtin_amdgcn_s_setprio(3);
    const char* wl = static_cast<const char*>(a.wpack) + (size_t)(wave * 2) * 1024 + lane * 16;
    // weight fragments of (chunk kc, step = tap * 4 + k-step); steps 36..38 are the first three of the NEXT unit's chunk
    // (the ring runs on across units; after the last unit they re-read its own, unused)
    auto wfrag = [&](int kc, int step, bf16x8 (&f)[2]) {
        const int over = step >= 36 ? 1 : 0;
        step -= 36 * over; kc ^= over;
        const int tap = step >> 2, ks = kc * 4 + (step & 3);
        const char* p = wl + (size_t)((tap * 8 + ks) * 16) * 1024;
        f[0] = *reinterpret_cast<const bf16x8*>(p);
        f[1] = *reinterpret_cast<const bf16x8*>(p + 1024);
    };
    auto toff_of = [&](int step) {
        step = min(step, 35);
        const int tap = step >> 2, ks = step & 3;
        return ((tap / 3 - 1) * kPW + (tap % 3 - 1)) * kStImgStride + ks * 64;
    };
    bf16x8 w0[2], w1[2], w2[2], w3[2];
    wfrag(0, 0, w0); wfrag(0, 1, w1); wfrag(0, 2, w2);
    f32x4 acc[MT][2];
    KA_LDS_BARRIER();                                        // unit 0 is staged
    const bool late = STAG && __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (late) KA_LDS_BARRIER();                              // waves 4-7 start one slot (half a unit) behind waves 0-3
    for (int u = 0; u < nunits; ++u) {
        const int bb = (int)blockIdx.x + (u >> 1) * nwg, kc = u & 1;
        if (!kc) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        int rowoff[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = mt * 16 + r;
            rowoff[mt] = (p < KA_BOARD ? (u & 1) * kPcImg + lds_square(0, p) * kStImgStride : kPcZero + (kPW + 1) * kStImgStride) + q * 16;
        }
        auto mm = [&](const bf16x8 (&wf)[2], const bf16x8 (&ac)[MT], bf16x8 (&an)[MT], int next_step) {
            const int toff = toff_of(next_step);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#ifndef KA_PC_NO_A
                an[mt] = *reinterpret_cast<const bf16x8*>(smem + rowoff[mt] + toff);
#else
                an[mt] = ac[mt]; (void)toff;                 // ablation build (tools/_diag/build_variants.sh): no activation-fragment LDS reads
#endif
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], ac[mt], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], ac[mt], acc[mt][1], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#ifndef KA_PC_NO_A
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        };
        bf16x8 fa[MT], fb[MT];
        {
            const int toff = toff_of(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(smem + rowoff[mt] + toff);
        }
#ifdef KA_PC_NO_W
#define wfrag(...) ((void)0)                                 // ablation build: the ring keeps its first fragments
        if (u == 0) w3[0] = w3[1] = w0[0];
#endif
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int s0 = tap * 4;
            // (the masked epilogue needs the ring's registers: before it the next unit's first fragments are not requested)
            const bool ahead = !(MASKED && kc && tap == 8);
            wfrag(kc, s0 + 3, w3); __builtin_amdgcn_sched_barrier(0); mm(w0, fa, fb, s0 + 1); __builtin_amdgcn_sched_barrier(0);
            if (ahead) wfrag(kc, s0 + 4, w0);
            __builtin_amdgcn_sched_barrier(0); mm(w1, fb, fa, s0 + 2); __builtin_amdgcn_sched_barrier(0);
            if (ahead) wfrag(kc, s0 + 5, w1);
            __builtin_amdgcn_sched_barrier(0); mm(w2, fa, fb, s0 + 3); __builtin_amdgcn_sched_barrier(0);
            if (ahead) wfrag(kc, s0 + 6, w2);
            __builtin_amdgcn_sched_barrier(0); mm(w3, fb, fa, s0 + 4); __builtin_amdgcn_sched_barrier(0);
            if (STAG && tap == 4) KA_LDS_BARRIER();          // slot boundary: 20 k-steps before it, 16 and the epilogue behind it
        }
#ifdef KA_PC_NO_W
#undef wfrag
#endif
        if (!MASKED && (u & 1) && !(a.tune_stagger & 1)) conv_epilogue<bf16_t, 2, MT>(a, acc, bb, wave * 2, 16, r, q);
        if (!(late && u == nunits - 1)) KA_LDS_BARRIER();    // this image may be overwritten, the next one is complete
        if (MASKED && kc) {
            // da = dh * [bn(y) > 0] and the BatchNorm-backward partial sums (conv_epilogue's masked branch, term for term), y from LDS
            // (opaque copies of the lane coordinates: everything below is invariant across the boards, and hoisted above the
            //  MFMA loop its addresses and coefficients spill it)
            int rl = r, ql = q;
            asm volatile("" : "+v"(rl), "+v"(ql));
            const int cb0 = chan_of(wave * 2, 4 * ql, 16);
            const int cbl = cb0;
            if (a.bsum) {                                    // per-board sums of the raw accumulators (conv_epilogue, same order)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bool in = mt * 16 + rl < KA_BOARD;
#pragma unroll
                        for (int i = 0; i < 4; ++i) s0[i] += in ? acc[mt][j][i] : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) s0[i] = row_sum16(s0[i]);
                    if (rl == 0) *reinterpret_cast<f32x4*>(a.bsum + (size_t)bb * a.Cout + cb0 + 4 * j) = f32x4{s0[0], s0[1], s0[2], s0[3]};
                }
            }
            // two passes of four channels (one MFMA tile each): half the coefficient / sum registers at a time; the first
            // pass's bf16 results wait in 12 registers so that the rows still leave as 16-byte pieces
            bf16x4 o0[MT];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float esc[4], esh[4], emu[4], eis[4], t1[4], t2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    esc[e] = a.ep_scale[cbl + 4 * j + e]; esh[e] = a.ep_shift[cbl + 4 * j + e];
                    emu[e] = a.ep_mean[cbl + 4 * j + e]; eis[e] = a.ep_invstd[cbl + 4 * j + e];
                    t1[e] = 0.f; t2[e] = 0.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = mt * 16 + rl;
                    const bool in = p < KA_BOARD;
                    bf16x4 yv = bf16x4{};
                    if (in) yv = *reinterpret_cast<const bf16x4*>(smem + kPcLds + p * kPcYStride + (cb0 + 4 * j) * 2);
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (float)yv[e];
                        const __bf16 db = (__bf16)acc[mt][j][e];
                        const float d = (in && y * esc[e] + esh[e] > 0.f) ? (float)db : 0.f;
                        t1[e] += d; t2[e] += d * ((y - emu[e]) * eis[e]);
                        o[e] = (__bf16)d;
                    }
                    if (j == 0) o0[mt] = o;
                    else if (in) {
                        bf16x8 o8;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o8[e] = o0[mt][e]; o8[4 + e] = o[e]; }
                        *reinterpret_cast<bf16x8*>(static_cast<char*>(a.out) + ((size_t)(bb * KA_BOARD + p) * a.Cout + cb0) * 2) = o8;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { t1[e] = row_sum16(t1[e]); t2[e] = row_sum16(t2[e]); }
                if (rl == 0) {
                    *reinterpret_cast<f32x4*>(a.ep_s1 + (size_t)bb * a.Cout + cb0 + 4 * j) = f32x4{t1[0], t1[1], t1[2], t1[3]};
                    *reinterpret_cast<f32x4*>(a.ep_s2 + (size_t)bb * a.Cout + cb0 + 4 * j) = f32x4{t2[0], t2[1], t2[2], t2[3]};
                }
            }
            wfrag(0, 0, w0); wfrag(0, 1, w1); wfrag(0, 2, w2);
        }
    }
}

template <bool TWO, bool MASKED, int NPW, int MT, bool STAG = false>
static int launch_conv_pc_form(const ConvArgs& a, int grid, size_t lds, hipStream_t st, const char* what) {
    static std::atomic<unsigned long long> done{0};          // per instantiation: devices already configured
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv3x3_pc_kernel<TWO, MASKED, NPW, MT, STAG>), done, what)) return rc;
    hipLaunchKernelGGL((conv3x3_pc_kernel<TWO, MASKED, NPW, MT, STAG>), dim3(grid), dim3(512 + 64 * NPW), lds, st, a);
    return ka_check_launch(what);
}
static int launch_conv_pc(ConvArgs a, hipStream_t st) {
    a.tune_stagger = 0; a.tune_prio = 0;                       // diagnostics: KA_CONV_P_ABL 1 no epilogue, 2 no staging after the first unit
    if (const char* e = ka_diag_env("KA_CONV_P_ABL")) a.tune_stagger = atoi(e);
    a.tune_prio = ka_opt(KA_OPT_CONV_P_PRIO, 0);
    int grid = 256;
    if (const int v = ka_opt(KA_OPT_CONV_P_WGS, 0); v > 0) grid = v;
    if (grid > a.B) grid = a.B;
    const int stag = ka_opt(KA_OPT_CONV_P_STAG, 0);            // KA_CONV_P_STAG=1: waves 4-7 half a unit behind waves 0-3 (five-row-tile forms)
#define KA_PC_FORM(TWO_, MASKED_, NPW_, LDS_, WHAT_)                                                         \
    (a.mt5 ? ((stag && !MASKED_) ? launch_conv_pc_form<TWO_, false, NPW_, 5, true>(a, grid, LDS_, st, WHAT_ ", 5 row tiles, staggered") \
                                 : launch_conv_pc_form<TWO_, MASKED_, NPW_, 5>(a, grid, LDS_, st, WHAT_ ", 5 row tiles"))            \
           : launch_conv_pc_form<TWO_, MASKED_, NPW_, kMTW>(a, grid, LDS_, st, WHAT_))
    if (a.in2 && a.ep_y) return KA_PC_FORM(true, true, 4, kPcLdsMasked, "conv3x3 (pc, masked)");
    if (a.in2) return KA_PC_FORM(true, false, 4, kPcLds, "conv3x3 (pc, two-tensor)");
    // two staging waves where the input carries a transform (its arithmetic shares the SIMDs with the MFMA waves: measured
    // 0.369-0.374 against 0.380-0.386 ms with four), four for the plain input (0.339 against 0.349); KA_CONV_P_NPW forces one
    if (ka_opt_set(KA_OPT_CONV_P_NPW) ? ka_opt(KA_OPT_CONV_P_NPW, 4) == 2 : (a.in_scale || a.relu || a.in_bias))
        return KA_PC_FORM(false, false, 2, kPcLds, "conv3x3 (pc, 2 staging waves)");
    return KA_PC_FORM(false, false, 4, kPcLds, "conv3x3 (pc)");
#undef KA_PC_FORM
}

// The masked data-gradient epilogue of a PAIR (conv3x3_pc2_kernel, index-order tiles: 0..4 the first board, 5..9 the second):
// da = dh * [bn(y) > 0] and the BatchNorm-backward partial sums (conv_epilogue's masked branch), one MFMA tile -- four channels per
// lane, 8-byte pieces -- at a time because the other board's accumulators stay live, and organised around what it costs: exposed
// latency and vector instructions that run while the matrix pipe idles (measured: the instruction count is what matters; the
// file is built with -fno-slp-vectorize, packed fp32 arithmetic beside MFMAs and its register pressure cost this epilogue 13 %):
//   * four stages (board, channel tile), the y pieces of stage s+1 requested before stage s is worked on, the first ones before
//     the per-board sums; the per-channel coefficients of a channel tile are loaded once for both boards;
//   * the second sum is taken as sum(d * y) and centred afterwards: S2 = (sum(d y) - mean * sum(d)) * invstd -- one fma per
//     element where (y - mean) * invstd * d took three instructions (equal up to fp32 rounding);
//   * bf16 <-> fp32 by shifts and masks on the packed words: the masked value d is the bf16 it is stored as.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int C, int MT>
__device__ __forceinline__ void conv_epilogue_masked_pair(const ConvArgs& a, f32x4 (&acc)[2 * MT][2], int b0, bool has_b1, int nt0, int r, int q) {
    constexpr int ROWB = C * 2, NT = C / 16;                  // bytes per square, 16-channel tiles
    const int cb[2] = {chan_of(nt0, 4 * q, NT), chan_of(nt0 + 1, 4 * q, NT)};
    const char* yb = static_cast<const char*>(a.ep_y) + (size_t)b0 * (KA_BOARD * ROWB);
    char* ob = static_cast<char*>(a.out) + (size_t)b0 * (KA_BOARD * ROWB);
    const int lo[2] = {r * ROWB + cb[0] * 2, r * ROWB + cb[1] * 2};
    // (MT = 6: the sixth row tile holds square 80 in lane row 0, fifteen padding rows behind it)
    auto live = [&](int mt) { return MT == 5 || mt < 5 || r == 0; };
    u32x2 yA[MT], yB[MT];
    auto yload = [&](int b, int j, u32x2 (&y)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) y[mt] = live(mt) ? *reinterpret_cast<const u32x2*>(yb + b * (KA_BOARD * ROWB) + lo[j] + mt * 16 * ROWB) : u32x2{0u, 0u};
    };
    f32x4 cA[2], cB[2];                                      // scale, shift of the channel tile's four channels
    yload(0, 0, yA);
    cA[0] = *reinterpret_cast<const f32x4*>(a.ep_scale + cb[0]); cA[1] = *reinterpret_cast<const f32x4*>(a.ep_shift + cb[0]);
    if (a.bsum) {                                            // per-board sums of the raw accumulators (conv_epilogue, same order)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s0[i] += live(mt) ? acc[MT * b + mt][j][i] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) s0[i] = row_sum16(s0[i]);
                if (r == 0 && (b == 0 || has_b1)) *reinterpret_cast<f32x4*>(a.bsum + (b0 + b) * C + cb[j]) = f32x4{s0[0], s0[1], s0[2], s0[3]};
            }
    }
    yload(0, 1, yB);
    cB[0] = *reinterpret_cast<const f32x4*>(a.ep_scale + cb[1]); cB[1] = *reinterpret_cast<const f32x4*>(a.ep_shift + cb[1]);
    auto process = [&](int b, int j, const u32x2 (&y)[MT], const f32x4 (&c)[2]) {
        const f32x4 emu = *reinterpret_cast<const f32x4*>(a.ep_mean + cb[j]), eis = *reinterpret_cast<const f32x4*>(a.ep_invstd + cb[j]);
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned yw = y[mt][h];
                const float y0 = __uint_as_float(yw << 16), y1 = __uint_as_float(yw & 0xffff0000u);
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 dbp = {(__bf16)acc[MT * b + mt][j][2 * h], (__bf16)acc[MT * b + mt][j][2 * h + 1]};
                const unsigned dw = __builtin_bit_cast(unsigned, dbp);
                const float d0 = (live(mt) && y0 * c[0][2 * h] + c[1][2 * h] > 0.f) ? __uint_as_float(dw << 16) : 0.f;
                const float d1 = (live(mt) && y1 * c[0][2 * h + 1] + c[1][2 * h + 1] > 0.f) ? __uint_as_float(dw & 0xffff0000u) : 0.f;
                t1[2 * h] += d0; t1[2 * h + 1] += d1;
                t2[2 * h] = fmaf(d0, y0, t2[2 * h]); t2[2 * h + 1] = fmaf(d1, y1, t2[2 * h + 1]);
                o[h] = (__float_as_uint(d0) >> 16) | __float_as_uint(d1);
            }
            if (live(mt)) *reinterpret_cast<u32x2*>(ob + b * (KA_BOARD * ROWB) + lo[j] + mt * 16 * ROWB) = o;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t1[e] = row_sum16(t1[e]);
            t2[e] = (row_sum16(t2[e]) - emu[e] * t1[e]) * eis[e];
        }
        if (r == 0) {
            *reinterpret_cast<f32x4*>(a.ep_s1 + (b0 + b) * C + cb[j]) = f32x4{t1[0], t1[1], t1[2], t1[3]};
            *reinterpret_cast<f32x4*>(a.ep_s2 + (b0 + b) * C + cb[j]) = f32x4{t2[0], t2[1], t2[2], t2[3]};
        }
    };
    process(0, 0, yA, cA);
    if (has_b1) yload(1, 0, yA);
    process(0, 1, yB, cB);
    if (has_b1) {
        yload(1, 1, yB);
        process(1, 0, yA, cA);
        process(1, 1, yB, cB);
    }
}

// The corner tile's epilogue inside conv3x3_pc2_kernel (ConvArgs::mt5 == 2): lane (r, q) holds, for channel tile j, the four
// channels cb[j] .. + 3 of square 80 of board bb.  conv3x3_corner_kernel's epilogue statement for statement -- the per-board sums
// the tower part of this kernel stored for squares 0..79 receive the corner's terms from the lane that owns (bb, cb[j]); they were
// stored by another lane of THIS wave (row 0 of the board's tiles): the caller waits for those stores, then the terms are added at the L2.
// Eight 16-byte reads straight from the L2 (sc1: device-coherent, past the CU's vector cache, which may hold a line from before this
// wave's own stores to it), requested together and complete when the statement ends -- ONE asm statement, so that the compiler
// cannot place a spill of a destination between its load and the wait (it does not know these loads are in flight).  What was
// measured instead: an acquire fence in front of plain loads costs 7-10 us per launch, fp32 atomic adds at the L2
// (global_atomic_add_f32, 16-32 per lane) 15-45 us.
__device__ __forceinline__ void ld4x8_l2(const float* const (&p)[8], f32x4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}
__device__ __forceinline__ void corner_tile_epilogue(const ConvArgs& a, const f32x4 (&acc)[2], int bb, bool live, int nt0, int q) {
    if (!live) return;
    int cb[2];
    f32x4 pbs[2], psq[2], ps1[2], ps2[2];
    {   // every sum this lane will add to, in one go (an array the launch does not carry reads the board's bsum / weight-pack line instead)
        const float* any = a.bsum ? a.bsum : reinterpret_cast<const float*>(a.wpack);
        const float* src[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cb[j] = chan_of(nt0 + j, 4 * q, 16);
            const size_t srow = (size_t)bb * 256 + cb[j];
            src[j] = a.bsum ? a.bsum + srow : any;
            src[2 + j] = a.sqpart ? a.sqpart + srow : any;
            src[4 + j] = a.ep_y ? a.ep_s1 + srow : any;
            src[6 + j] = a.ep_y ? a.ep_s2 + srow : any;
        }
        f32x4 got[8];
        ld4x8_l2(src, got);
#pragma unroll
        for (int j = 0; j < 2; ++j) { pbs[j] = got[j]; psq[j] = got[2 + j]; ps1[j] = got[4 + j]; ps2[j] = got[6 + j]; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x4 v = acc[j];
        const size_t orow = ((size_t)bb * KA_BOARD + 80) * 256 + cb[j], srow = (size_t)bb * 256 + cb[j];
        if (a.bsum) *reinterpret_cast<f32x4*>(a.bsum + srow) = pbs[j] + v;
        if (a.sqpart) *reinterpret_cast<f32x4*>(a.sqpart + srow) = psq[j] + f32x4{v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]};
        bf16x4 o;
        if (!a.ep_y) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        } else {
            // da = dh * [bn(y) > 0] and its BatchNorm-backward terms (conv_epilogue's masked branch, term for term)
            const f32x4 esc = *reinterpret_cast<const f32x4*>(a.ep_scale + cb[j]), esh = *reinterpret_cast<const f32x4*>(a.ep_shift + cb[j]);
            const f32x4 emu = *reinterpret_cast<const f32x4*>(a.ep_mean + cb[j]), eis = *reinterpret_cast<const f32x4*>(a.ep_invstd + cb[j]);
            const bf16x4 yv = *reinterpret_cast<const bf16x4*>(static_cast<const char*>(a.ep_y) + orow * 2);
            f32x4 t1, t2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (float)yv[e];
                const __bf16 db = (__bf16)v[e];
                const float d = (y * esc[e] + esh[e] > 0.f) ? (float)db : 0.f;
                t1[e] = d; t2[e] = d * ((y - emu[e]) * eis[e]);
                o[e] = (__bf16)d;
            }
            *reinterpret_cast<f32x4*>(a.ep_s1 + srow) = ps1[j] + t1;
            *reinterpret_cast<f32x4*>(a.ep_s2 + srow) = ps2[j] + t2;
        }
        *reinterpret_cast<bf16x4*>(static_cast<char*>(a.out) + orow * 2) = o;
    }
}

// ---------------------------------------------------------------- two boards per weight fragment (bf16, Cin = Cout = 256)
// What paces conv3x3_pc_kernel is not the matrix pipe but the weight stream: every (board, chunk) unit pulls its 590 KB of
// fragments out of the L2 again -- 4.8 GB per launch at B = 4096, i.e. 15 TB/s through a path that delivers about 70 GB/s per
// CU (TA busy 68 %, TCP pending-stall 38 % of the kernel's cycles at a matrix pipe 61 % busy: profiles/r04_conv_vmem_path_counters.json).
// Here a unit is TWO boards x a 64-channel chunk: an MFMA wave owns ten row tiles (squares 0..79 of both boards) x two channel
// tiles, so a weight fragment feeds 20 MFMAs instead of 10 and the stream per board is halved; the four 64-channel images
// (two boards x two buffers) take the LDS the two 128-channel images took.  Activation fragments are read one k-step ahead into
// the registers of the tile just multiplied (one set, no second buffer); every LDS offset of the 18 k-steps of a unit is an
// immediate.  Same weight packs, staging transforms and epilogue as conv3x3_pc_kernel; the k-steps of an output element are
// summed in the order (64-channel chunk, tap, k-step) instead of (128-channel chunk, tap, k-step): results equal those of the
// other kernels up to fp32 re-association.
constexpr int kP2Stride = 64 * 2 + 32;                                  // bytes per square of a 64-channel image: 16 fragment lanes on 16 bank slots
constexpr int kP2Img = kImgSquares1 * kP2Stride;                        // 28 960 B
constexpr int kP2Lds = 4 * kP2Img;                                      // [buffer][board]
// in-kernel corner (ConvArgs::mt5 == 2): behind the images, nine pair slots x two boards of [4 corner squares][256 channels] bf16 --
// the rows of the corner tile -- and a 16-byte dump slot for the staging lanes that hold no corner square
constexpr int kP2SideRow = 4 * 512 + 16;                                // as kCornerStride: 16 fragment lanes on 16 bank slots
constexpr int kP2SideSlots = 9;
constexpr int kP2Side = kP2Lds, kP2Dump = kP2Side + 2 * kP2SideSlots * kP2SideRow, kP2LdsCorner = kP2Dump + 16;
constexpr int kP2Base = (kPW + 1) * kP2Stride;                          // the most negative tap offset, folded into the row base

// ---- border tiles (SKIP).  A tap that steps off the board multiplies the zero halo: 104 of the 729 (square, tap) products of a
// board.  With squares taken in index order no row tile is all halo for any tap; with two boards per unit the rows can be dealt so
// that three tiles are: TOP = squares (0, 0..7) of both boards (zero for the three taps with dy = -1), RIGHT = (0..7, 8) (dx = +1),
// LEFT = (1..8, 0) (dx = -1).  The fourth border tile holds (8, 1..7) and one interior square (4, 4) of each board -- square
// (8, 8) stays with conv3x3_corner_kernel -- and the 48 other interior squares of a board are its three interior tiles.  Nine of
// the 90 (tile, tap) pairs of a k-step pair are skipped: 10 % of the MFMAs and of the activation-fragment reads.
// Bank conflicts: a 16-lane ds_read_b128 group is conflict-free when the squares of lanes r in {0..3, 12..15} have distinct LDS
// indices mod 8, and those of lanes 4..11 too (a square is 160 B = 10 sixteen-byte slots, and 17 = 1 mod 8 squares per padded
// row): index mod 8 = (x + y + 2) mod 8, so each half tile takes squares with eight different (x + y) mod 8 -- an edge is
// such a set, (8, 1..7) + (4, 4) is, and the 48 interior squares split into six (every residue occurs exactly six times).
// Lanes {0..3, 12..15} of a border tile belong to the pair's first board, lanes 4..11 to the second.
struct P2Perm { unsigned char sq[10][16]; };
constexpr P2Perm make_p2perm() {
    P2Perm P{};
    const int E[8] = {0, 1, 2, 3, 12, 13, 14, 15}, O[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    for (int k = 0; k < 8; ++k) {
        const int edge[4] = {k, 9 * k + 8, 9 * (k + 1), k < 7 ? 72 + (k + 1) : 40};
        for (int t = 0; t < 4; ++t) { P.sq[t][E[k]] = (unsigned char)edge[t]; P.sq[t][O[k]] = (unsigned char)edge[t]; }
    }
    int byres[8][6] = {}, cnt[8] = {};
    for (int y = 1; y <= 7; ++y)
        for (int x = 1; x <= 7; ++x) {
            if (y == 4 && x == 4) continue;
            const int rho = (x + y) & 7;
            byres[rho][cnt[rho]++] = 9 * y + x;
        }
    for (int b = 0; b < 2; ++b)
        for (int tt = 0; tt < 3; ++tt)
            for (int k = 0; k < 8; ++k) {
                P.sq[4 + 3 * b + tt][E[k]] = (unsigned char)byres[k][2 * tt];
                P.sq[4 + 3 * b + tt][O[k]] = (unsigned char)byres[k][2 * tt + 1];
            }
    return P;
}
__device__ const P2Perm kP2Perm = make_p2perm();
__host__ __device__ constexpr bool p2_skip(int t, int tap) {
    return (t == 0 && tap / 3 == 0) || (t == 1 && tap % 3 == 2) || (t == 2 && tap % 3 == 0);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) -- every index a constant expression, so
// arrays indexed by table entries stay in registers
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// the (k-step, tile) pairs of a unit in issue order, without the skipped ones
struct P2Seq { unsigned char s[216], t[216]; int n; };
constexpr P2Seq make_p2seq(bool skip, int ntiles) {
    P2Seq Q{};
    Q.n = 0;
    for (int s = 0; s < 18; ++s)
        for (int t = 0; t < ntiles; ++t)
            if (!(skip && p2_skip(t, s >> 1))) { Q.s[Q.n] = (unsigned char)s; Q.t[Q.n] = (unsigned char)t; ++Q.n; }
    return Q;
}

// register-only epilogue of a pair in the border-tile layout: lane (r, q) holds, for tile t, square sqv[t] of board
// b0 + (t < 4 ? lanes 4..11 : t >= 7) and for channel tile j the 4 channels cb[j]..+3 (both tiles: 8 consecutive channels)
__device__ __forceinline__ void conv_epilogue_pair(const ConvArgs& a, f32x4 (&acc)[10][2], const unsigned (&sqp)[3], int b0, bool has_b1,
                                                   int nt0, int r, int q) {       // sqp: the lane's ten squares, one byte each
    const int cb[2] = {chan_of(nt0, 4 * q, 16), chan_of(nt0 + 1, 4 * q, 16)};
    const bool isO = r >= 4 && r < 12;
    if (a.bsum || a.sqpart) {
        // one (channel tile, board) at a time: eight running sums live beside the eighty accumulators
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    if (t >= 4 && (t >= 7) != (b == 1)) continue;            // an interior tile of the other board
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[t][j][i];
                        if (t < 4) v = (isO == (b == 1)) ? v : 0.f;          // border tile: this lane's row belongs to one board
                        s[i] += v; ss[i] += v * v;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { s[i] = row_sum16(s[i]); ss[i] = row_sum16(ss[i]); }
                if (r == 0 && (b == 0 || has_b1)) {
                    if (a.bsum) *reinterpret_cast<f32x4*>(a.bsum + (size_t)(b0 + b) * 256 + cb[j]) = f32x4{s[0], s[1], s[2], s[3]};
                    if (a.sqpart) *reinterpret_cast<f32x4*>(a.sqpart + (size_t)(b0 + b) * 256 + cb[j]) = f32x4{ss[0], ss[1], ss[2], ss[3]};
                }
                __builtin_amdgcn_sched_barrier(0);           // (one block of sums at a time: interleaved they spill the MFMA loop)
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const int bd = t < 4 ? (isO ? 1 : 0) : (t < 7 ? 0 : 1);
        if (bd && !has_b1) continue;
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = (__bf16)acc[t][0][i]; o[4 + i] = (__bf16)acc[t][1][i]; }
        const int sq = (int)((sqp[t >> 2] >> (8 * (t & 3))) & 0xffu);
        // (one uniform base per pair and a 32-bit lane offset)
        *reinterpret_cast<bf16x8*>(static_cast<char*>(a.out) + (size_t)b0 * (KA_BOARD * 512) + ((bd * KA_BOARD + sq) * 256 + cb[0]) * 2) = o;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// STAG: MFMA waves 4-7 (the SIMD partners of waves 0-3) run half a unit (nine k-steps) behind waves 0-3, two barriers per unit, so
// that one partner's epilogue and unit-start latencies sit beside the other's MFMAs; an image pair is then read for three
// half-unit slots, and the staging waves load the next unit in one slot and write it in the following one (the schedule of
// conv3x3_pc_kernel<..., STAG>, which did not pay there: that kernel waits for its weight stream, this one for the matrix pipe).
// C = 256, MT = 5: eight MFMA waves, four 64-channel chunks, squares 0..79 (square 80: conv3x3_corner_kernel).  C = 128, MT = 6: four
// MFMA waves (one per SIMD, 256 registers each beside the staging waves), two chunks, all 81 squares as six row tiles per board.
template <int C, int MT, bool TWO, bool MASKED, int NPW, bool SKIP = false, bool STAG = false, bool GATED = false>
__global__ __launch_bounds__((C / 32 + NPW) * 64) void conv3x3_pc2_kernel(ConvArgs a) {
    static_assert(!GATED || (TWO && MASKED && !SKIP && !STAG), "the gated input is conv2's data gradient: two-tensor, masked");
    static_assert(!(SKIP && MASKED), "the border-tile layout is built for the register-only epilogue");
    static_assert((C == 256 && MT == 5) || (C == 128 && MT == 6 && !SKIP), "shapes this kernel is built for");
    constexpr int NMW = C / 32, NCH = C / 64, NTILE = C / 16, KSG = C / 32, ROWB = C * 2, NTL = 2 * MT;
    constexpr int NT_ = (NMW + NPW) * 64, NP = 64 * NPW;
    constexpr int kHalf = KA_BOARD * 8, kPieces = 2 * kHalf;            // 16-byte pieces of a unit: 2 boards x 81 squares x 8
    constexpr int KP = (kPieces + NP - 1) / NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x, npairs_all = (a.B + 1) >> 1;
    if ((int)blockIdx.x >= npairs_all) return;
    for (int i = tid; i < kP2Lds / 16; i += NT_) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
    const int npairs = (npairs_all - (int)blockIdx.x + nwg - 1) / nwg, nunits = NCH * npairs;
    if (!MASKED) a.ep_y = nullptr;                           // (compile-time: conv_epilogue's masked branch is not compiled in)
    __syncthreads();

    if (wave >= NMW) {
        // ---------------- staging waves: piece i = pt + NP k of a unit = row i / 8 of the pair's 2 x 81 rows, 16-byte piece i % 8.
        // Everything a piece needs besides its data is a lane constant (its LDS slot, its byte offset inside the pair) or scalar
        // (the pair, the chunk): the tensors are addressed through buffer descriptors sized to the batch, so the rows of a missing
        // second board read as zeros and are never written, without a branch.  The vector instructions of these waves share the
        // SIMDs with the MFMA waves one for one.
        const int pt = tid - NMW * 64, pc = pt & 7, swave = __builtin_amdgcn_readfirstlane(wave) - NMW;
        const bool has_aff = GATED || a.in_scale != nullptr;
        const unsigned nbytes = (unsigned)a.B * (KA_BOARD * ROWB);
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(TWO ? a.in2 : a.in), 0, nbytes, 0x00020000);
        // (in_out: the two-tensor form's dy; the forward transform form's x' when the caller keeps it for the weight gradient --
        //  a null in_out is a descriptor of zero bytes: the stores are dropped)
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.in_out ? a.in_out : const_cast<void*>(a.in), 0,
                                                                              a.in_out ? nbytes : 0u, 0x00020000);
        const int voff0 = (pt >> 3) * ROWB + pc * 16;
        int ldso[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int row = min((pt + NP * k) >> 3, 2 * KA_BOARD - 1), j = row >= KA_BOARD ? 1 : 0;
            ldso[k] = j * kP2Img + lds_square(0, row - j * KA_BOARD) * kP2Stride + pc * 16;
        }
        // round k of this wave has pieces at all?  (NP = 256: the sixth round is sixteen lanes of the first staging wave)
        auto live = [&](int k) { return (swave * 64 + NP * k) < kPieces; };
        // in-kernel corner: a piece of squares 70, 71, 79, 80 (taps 0, 1, 3, 4 of square 80) is ALSO written, transformed as it is,
        // into the pair's slot of the side buffer; cso[k]: its offset inside the slot, -1 for every other piece (which writes the
        // dump slot instead: the store itself stays unconditional)
        const bool corner_in = C == 256 && MT == 5 && !MASKED && a.mt5 == 2;
        int cso[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int row = min((pt + NP * k) >> 3, 2 * KA_BOARD - 1), j = row >= KA_BOARD ? 1 : 0, sq = row - j * KA_BOARD;
            const int ps = sq == 70 ? 0 : sq == 71 ? 1 : sq == 79 ? 2 : sq == 80 ? 3 : -1;
            cso[k] = (corner_in && ps >= 0 && pt + NP * k < kPieces) ? j * kP2SideRow + ps * 512 + pc * 16 : -1;
        }
        int chas = 0;                                            // bit k: some lane of this wave holds a corner piece in round k
#pragma unroll
        for (int k = 0; k < KP; ++k) chas |= (__builtin_amdgcn_readfirstlane((int)(__ballot(cso[k] >= 0) != 0ull)) & 1) << k;
        bf16x8 pv[KP], pw[TWO ? KP : 1];
        auto stage_load = [&](int u) {
            const int b0 = 2 * ((int)blockIdx.x + (u / NCH) * nwg), c4 = u % NCH;
            const int soff = __builtin_amdgcn_readfirstlane(b0 * (KA_BOARD * ROWB) + c4 * 128);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (!live(k)) continue;
                const int vo = pt + NP * k < kPieces ? voff0 + k * (NP / 8) * ROWB : (int)0x7fffffff;      // (past the pair: out of range)
                pv[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_in, vo, soff, 2));
                if (TWO) pw[TWO ? k : 0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_in2, vo, soff, 2));
            }
        };
        auto stage_write = [&](int u) __attribute__((always_inline)) {
            const int b0 = 2 * ((int)blockIdx.x + (u / NCH) * nwg), c4 = u % NCH, ch0 = c4 * 64 + pc * 8;
            const int soff = __builtin_amdgcn_readfirstlane(b0 * (KA_BOARD * ROWB) + c4 * 128);
            char* img = smem + (u & 1) * (2 * kP2Img);
            float sc[8], sh[8], k3[8], pb[2][8];
            {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f}, o = {1.f, 1.f, 1.f, 1.f};
                // (GATED: the three coefficient vectors are one array k[3][C] and gate / add one array [2][B][C] -- two base pointers in
                //  scalar registers instead of five: with five the weight descriptor no longer fits the scalar file, moves to vector
                //  registers and every weight load of the MFMA loop becomes a readfirstlane loop)
                const float* shp = GATED ? a.in_scale + C : a.in_shift;
                const float* k3p = GATED ? a.in_scale + 2 * C : a.in_k3;
                const f32x4 s0 = has_aff ? *reinterpret_cast<const f32x4*>(a.in_scale + ch0) : o, s1 = has_aff ? *reinterpret_cast<const f32x4*>(a.in_scale + ch0 + 4) : o;
                const f32x4 t0 = has_aff ? *reinterpret_cast<const f32x4*>(shp + ch0) : z, t1 = has_aff ? *reinterpret_cast<const f32x4*>(shp + ch0 + 4) : z;
                const f32x4 u0 = TWO ? *reinterpret_cast<const f32x4*>(k3p + ch0) : z, u1 = TWO ? *reinterpret_cast<const f32x4*>(k3p + ch0 + 4) : z;
                const bool bias = !TWO && a.in_bias;
                const f32x4 p00 = bias ? *reinterpret_cast<const f32x4*>(a.in_bias + (size_t)b0 * C + ch0) : z;
                const f32x4 p01 = bias ? *reinterpret_cast<const f32x4*>(a.in_bias + (size_t)b0 * C + ch0 + 4) : z;
                const bool b1 = bias && b0 + 1 < a.B;
                const f32x4 p10 = b1 ? *reinterpret_cast<const f32x4*>(a.in_bias + (size_t)(b0 + 1) * C + ch0) : z;
                const f32x4 p11 = b1 ? *reinterpret_cast<const f32x4*>(a.in_bias + (size_t)(b0 + 1) * C + ch0 + 4) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = t0[e]; sh[4 + e] = t1[e]; k3[e] = u0[e]; k3[4 + e] = u1[e];
                    pb[0][e] = p00[e]; pb[0][4 + e] = p01[e]; pb[1][e] = p10[e]; pb[1][4 + e] = p11[e];
                }
            }
            // gated two-tensor form: per board j, (in*gate_j + add_j)*sc + sh = in*(gate_j*sc) + (add_j*sc + sh)
            float gsc0[8], gsh0[8], gsc1[8], gsh1[8];
            if constexpr (GATED) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const float* gp = a.in_bias + (size_t)b0 * C + ch0;
                const float* ap = gp + (size_t)a.B * C;
                const bool b1 = b0 + 1 < a.B;
                const f32x4 g00 = *reinterpret_cast<const f32x4*>(gp), g01 = *reinterpret_cast<const f32x4*>(gp + 4);
                const f32x4 a00 = *reinterpret_cast<const f32x4*>(ap), a01 = *reinterpret_cast<const f32x4*>(ap + 4);
                const f32x4 g10 = b1 ? *reinterpret_cast<const f32x4*>(gp + C) : z, g11 = b1 ? *reinterpret_cast<const f32x4*>(gp + C + 4) : z;
                const f32x4 a10 = b1 ? *reinterpret_cast<const f32x4*>(ap + C) : z, a11 = b1 ? *reinterpret_cast<const f32x4*>(ap + C + 4) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gsc0[e] = g00[e] * sc[e]; gsc0[4 + e] = g01[e] * sc[4 + e];
                    gsh0[e] = fmaf(a00[e], sc[e], sh[e]); gsh0[4 + e] = fmaf(a01[e], sc[4 + e], sh[4 + e]);
                    gsc1[e] = g10[e] * sc[e]; gsc1[4 + e] = g11[e] * sc[4 + e];
                    gsh1[e] = fmaf(a10[e], sc[e], sh[e]); gsh1[4 + e] = fmaf(a11[e], sc[4 + e], sh[4 + e]);
                }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (!live(k)) continue;
                const int i = pt + NP * k;
                if (i >= kPieces) continue;
                const int j = i >= kHalf ? 1 : 0;
                bf16x8 v = pv[k];
                if (TWO) {
                    // (the board of a piece is a compile-time fact in every round but the one the pair's middle falls in)
                    const bool jlo = NP * (k + 1) <= kHalf, jhi = NP * k >= kHalf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float se_ = sc[e], he_ = sh[e];
                        if constexpr (GATED) {
                            const float c0 = gsc0[e], c1 = gsc1[e], h0 = gsh0[e], h1 = gsh1[e];
                            se_ = jlo ? c0 : jhi ? c1 : (j ? c1 : c0);
                            he_ = jlo ? h0 : jhi ? h1 : (j ? h1 : h0);
                        }
                        v[e] = (__bf16)fmaf((float)pw[TWO ? k : 0][e], k3[e], fmaf((float)v[e], se_, he_));
                    }
                    if (b0 + j >= a.B) v = bf16x8{};          // (a missing board stays all zeros: the transform of zeros is the shift)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r_out,
                                                           voff0 + k * (NP / 8) * ROWB, soff, 0);
                } else if (has_aff || a.relu || a.in_bias) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 f = {(float)v[e], (float)v[e + 1]};
                        if (has_aff) f = __builtin_elementwise_fma(f, f32x2{sc[e], sc[e + 1]}, f32x2{sh[e], sh[e + 1]});
                        if (a.relu) f = __builtin_elementwise_max(f, f32x2{0.f, 0.f});
                        if (a.in_bias) f += j ? f32x2{pb[1][e], pb[1][e + 1]} : f32x2{pb[0][e], pb[0][e + 1]};
                        v[e] = (__bf16)f[0]; v[e + 1] = (__bf16)f[1];
                    }
                    if (b0 + j >= a.B) v = bf16x8{};
                    // (the transformed input, kept for the weight gradient when the caller asked: ka_conv3x3_fwd_keep.  256 channels
                    //  only: in the 128-channel kernel this store makes the compiler spill 656 bytes per lane)
                    if constexpr (C == 256) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r_out,
                                                           voff0 + k * (NP / 8) * ROWB, soff, 0);
                }
                *reinterpret_cast<bf16x8*>(img + ldso[k]) = v;
            }
        };
        // in-kernel corner: the lanes that just wrote a piece of a corner square copy it (their own LDS write, read back in order) into
        // the pair's slot of the side buffer -- pair (u / NCH) of this workgroup, nine slots in turn, the chunk's 128 bytes inside a
        // square; every other lane moves 16 bytes of its piece to the dump slot.  (A loop of its own: a second store per piece inside
        // stage_write makes the compiler spill ~900 bytes per lane across the whole kernel.)
        auto side_write = [&](int u) {
            if constexpr (C == 256 && MT == 5) {
                if (!corner_in) return;
                const char* img = smem + (u & 1) * (2 * kP2Img);
                const int side0 = kP2Side + 2 * ((u / NCH) % kP2SideSlots) * kP2SideRow + (u % NCH) * 128;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    if (!live(k) || !((chas >> k) & 1)) continue;        // (wave-uniform: this wave holds a corner piece in round k)
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(img + ldso[k]);
                    *reinterpret_cast<bf16x8*>(smem + (cso[k] >= 0 ? side0 + cso[k] : kP2Dump)) = t;
                }
            }
        };
        auto stage = [&](int u) __attribute__((always_inline)) { stage_load(u); stage_write(u); side_write(u); };   // (inlined by force: as a call its closure -- and the kernel arguments -- live in scratch)
        stage(0);
        KA_LDS_BARRIER();
        if constexpr (STAG) {
            // slot 2v-2: the loads of unit v; slot 2v-1: its LDS writes (waves 4-7 read that buffer's previous unit until then)
            for (int v = 1; v < nunits; ++v) {
                stage_load(v);
                KA_LDS_BARRIER();
                stage_write(v);
                side_write(v);
                KA_LDS_BARRIER();
            }
            KA_LDS_BARRIER();
            KA_LDS_BARRIER();
            return;
        }
        for (int u = 0; u < nunits; ++u) {
            if (u + 1 < nunits) stage(u + 1);
            KA_LDS_BARRIER();
        }
        return;
    }

    // ---------------- MFMA waves: ten row tiles (five per board) x two channel tiles; the weight ring (three slots, two k-steps
    // ahead: a k-step is 20 MFMAs per wave) runs on across the units
    // weight fragments of step s of 64-channel chunk c4: s = tap * 2 + k-step; steps 18, 19 are the first two of the next unit's chunk.
    // The address is a scalar base (the wave's channel tiles, the step) plus the lane's 32-bit offset: the step walk stays on the
    // scalar unit -- as 64-bit lane pointers it cost two vector adds and two registers per load pair
    // (buffer loads: descriptor of the whole pack in four scalar registers, the lane's 32-bit offset, the step as the scalar offset)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpack), 0, 9 * KSG * NTILE * 1024, 0x00020000);
    const int wlane = __builtin_amdgcn_readfirstlane(wave) * 2048 + lane * 16;
    auto wfrag = [&](int c4, int s, bf16x8 (&f)[2]) {
        if (s >= 18) { s -= 18; c4 = (c4 + 1) % NCH; }
        const int tap = s >> 1, ks = c4 * 2 + (s & 1);
        // (readfirstlane: the unit counter is wave-uniform, but it lives under the wave-role branch and is not provably so)
        const int so = __builtin_amdgcn_readfirstlane(((tap * KSG + ks) * NTILE) * 1024);
        f[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        f[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0));
    };
    // LDS byte offset of (row tile t, lane) in the first buffer, minus kP2Base.  Index order: tile t of board 0 (board 1: + kP2Img);
    // SKIP: ten entries, the lane's (board, square) of every tile of the border-tile layout
    constexpr int NRB = SKIP ? 10 : MT;
    int rowbase[NRB];
    unsigned sqp[3] = {0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NRB; ++t) {
        if (SKIP) {
            const int sq = kP2Perm.sq[t][r];
            sqp[t >> 2] |= (unsigned)sq << (8 * (t & 3));
            const int bd = t < 4 ? ((r >= 4 && r < 12) ? 1 : 0) : (t < 7 ? 0 : 1);
            rowbase[t] = bd * kP2Img + lds_square(0, sq) * kP2Stride + q * 16 - kP2Base;
        } else {
            // (MT = 6: the padding rows 81..95 read square index kPW + 13 of the image: it and its eight neighbours are padding
            //  squares to the right of the board, never written, all zero)
            const int p = t * 16 + r;
            rowbase[t] = (p < KA_BOARD ? lds_square(0, p) : kPW + 13) * kP2Stride + q * 16 - kP2Base;
        }
    }
    // (rowbase follows the buffer: + 2 kP2Img for the odd units, toggled in place at every unit's end -- a second set of ten offsets
    //  for the other buffer does not fit the register budget)
    auto frag_off = [&](int t) { return SKIP ? rowbase[SKIP ? t : 0] : rowbase[t % MT] + (t / MT) * kP2Img; };
    bf16x8 wr[3][2];
    wfrag(0, 0, wr[0]); wfrag(0, 1, wr[1]);
    f32x4 acc[NTL][2];
    KA_LDS_BARRIER();                                        // unit 0 is staged
    if (a.stamps && tid == 0) {                              // diagnostic only (ka_debug_conv_stamps; null in production): the clock the kernel held
        a.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
        a.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    const bool late = STAG && __builtin_amdgcn_readfirstlane(wave) >= NMW / 2;
    if (late) KA_LDS_BARRIER();                              // waves 4-7 start one slot (half a unit) behind waves 0-3
    for (int u = 0; u < nunits; ++u) {
        const int c4 = u % NCH;
        if (!c4) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        const char* img = smem;
        // activation fragments: a ring of R, walked over the unit's (k-step, tile) pairs in issue order; each slot is refilled right
        // after the two MFMAs that read it with the fragment R pairs further on (R = 8: 16 MFMAs of this wave, 256-512 cycles
        // of the shared pipe, ahead of its use)
        constexpr P2Seq Q = make_p2seq(SKIP, NTL);
        constexpr int R = MASKED ? 5 : 8;
        bf16x8 fa[R];
        auto frag_at = [&](auto n_) {
            constexpr int n = decltype(n_)::value, s = Q.s[n], t = Q.t[n], tap = s >> 1;
            constexpr int toff = kP2Base + ((tap / 3 - 1) * kPW + (tap % 3 - 1)) * kP2Stride + (s & 1) * 64;
            fa[n % R] = *reinterpret_cast<const bf16x8*>(img + frag_off(t) + toff);
        };
        static_for<R>(frag_at);
        static_for<Q.n>([&](auto n_) {
            constexpr int n = decltype(n_)::value, s = Q.s[n], t = Q.t[n];
            if constexpr (n == 0 || Q.s[n > 0 ? n - 1 : 0] != s) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STAG && s == 9) KA_LDS_BARRIER();      // slot boundary: nine k-steps before it, nine and the epilogue behind it
                // (the masked epilogue needs the ring's registers: before it the next unit's first fragments are not requested)
                if (!(MASKED && c4 == NCH - 1 && s >= 16)) wfrag(c4, s + 2, wr[(s + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s % 3][0], fa[n % R], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s % 3][1], fa[n % R], acc[t][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if constexpr (n + R < Q.n) {
                frag_at(std::integral_constant<int, n + R>{});
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if (c4 == NCH - 1) {
            const int b0 = 2 * ((int)blockIdx.x + (u / NCH) * nwg);
            if constexpr (MASKED) {
                // (opaque copies of the lane coordinates: the epilogue's addresses are invariant across the pairs, and hoisted
                //  above the MFMA loop they spill it)
                int rl = r, ql = q;
                asm volatile("" : "+v"(rl), "+v"(ql));
                conv_epilogue_masked_pair<C, MT>(a, acc, b0, b0 + 1 < a.B, wave * 2, rl, ql);
                wfrag(0, 0, wr[0]); wfrag(0, 1, wr[1]);
            } else if constexpr (SKIP) {
                if constexpr (SKIP) conv_epilogue_pair(a, acc, sqp, b0, b0 + 1 < a.B, wave * 2, r, q);
            } else {
                conv_epilogue<bf16_t, 2, MT>(a, reinterpret_cast<f32x4 (&)[MT][2]>(acc[0]), b0, wave * 2, NTILE, r, q);
                if (b0 + 1 < a.B) conv_epilogue<bf16_t, 2, MT>(a, reinterpret_cast<f32x4 (&)[MT][2]>(acc[MT]), b0 + 1, wave * 2, NTILE, r, q);
            }
            if constexpr (C == 256 && MT == 5 && !SKIP && !STAG && !MASKED) {
                // ---- in-kernel corner (mt5 == 2; not compiled into the masked form: there it measured 6 us slower than the launch,
                // and its registers cost that form another 52 bytes of scratch per lane): square 80 of the boards of up to eight pairs is ONE more row tile.  Its rows --
                // taps 0, 1, 3, 4 of the square, 256 channels, transformed -- were left in the side buffer by the staging waves,
                // which at this point wait at the unit's barrier (the pair they staged last went into the ninth slot).  Every MFMA
                // wave pulls the tile's 64 weight fragments twelve-less-four steps ahead and runs 64 MFMAs in the order of
                // conv3x3_corner_kernel, (128-channel chunk, tap, k-step): out[b, 80, :] and the sums are bit-identical to that kernel's.
                const int pidx = u / NCH;
                if (a.mt5 == 2 && ((pidx & 7) == 7 || pidx == npairs - 1)) {
                    const int g0 = pidx & ~7;
                    const int pr = g0 + (r >> 1), prc = min(pr, pidx);
                    const char* arow = smem + kP2Side + (2 * (prc % kP2SideSlots) + (r & 1)) * kP2SideRow + q * 16;
                    auto cw = [&](int step, bf16x8 (&f)[2]) {
                        const int ps = (step >> 2) & 3, tap = ps < 2 ? ps : ps + 1, ks = (step >> 4) * 4 + (step & 3);
                        const int so = __builtin_amdgcn_readfirstlane(((tap * KSG + ks) * NTILE) * 1024);
                        f[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
                        f[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0));
                    };
                    constexpr int kAh = 8;
                    bf16x8 wq[kAh][2];
#pragma unroll
                    for (int s2 = 0; s2 < kAh; ++s2) cw(s2, wq[s2]);
                    f32x4 cacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int step = 0; step < 32; ++step) {
                        const bf16x8 w0 = wq[step % kAh][0], w1 = wq[step % kAh][1];
                        if (step + kAh < 32) cw(step + kAh, wq[step % kAh]);
                        const int ps = (step >> 2) & 3, ks = (step >> 4) * 4 + (step & 3);
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + ps * 512 + ks * 64);
                        cacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af, cacc[0], 0, 0, 0);
                        cacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af, cacc[1], 0, 0, 0);
                    }
                    const int bb = 2 * ((int)blockIdx.x + pr * nwg) + (r & 1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the sums of squares 0..79 this wave stored have reached the L2
                    corner_tile_epilogue(a, cacc, bb, pr <= pidx && bb < a.B, wave * 2, q);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NRB; ++t) rowbase[t] += (u & 1) ? -2 * kP2Img : 2 * kP2Img;
        if (!(late && u == nunits - 1)) KA_LDS_BARRIER();    // these images may be overwritten, the next pair is complete
    }
    if (a.stamps && tid == 0) {
        a.stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
        a.stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int C, int MT, bool TWO, bool MASKED, int NPW, bool SKIP = false, bool STAG = false, bool GATED = false>
static int launch_conv_pc2_form(const ConvArgs& a, hipStream_t st, const char* what) {
    static std::atomic<unsigned long long> done{0};          // per instantiation: devices already configured
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv3x3_pc2_kernel<C, MT, TWO, MASKED, NPW, SKIP, STAG, GATED>), done, what)) return rc;
    const int pairs = (a.B + 1) / 2, grid = pairs < 256 ? pairs : 256;
    hipLaunchKernelGGL((conv3x3_pc2_kernel<C, MT, TWO, MASKED, NPW, SKIP, STAG, GATED>), dim3(grid), dim3((C / 32 + NPW) * 64),
                       C == 256 ? kP2LdsCorner : kP2Lds, st, a);       // (the 256-channel forms always carry the side buffer: their staging stores are unconditional)
    return ka_check_launch(what);
}
static int launch_conv_pc2(const ConvArgs& a, hipStream_t st) {
    if (a.in2 && a.in_bias) {                                  // conv2's data gradient taking (du, gate, add) for dz: ka_conv3x3_dgrad_fused_gated
        KA_REQUIRE(a.in2 && a.ep_y, "conv3x3: the gated input comes with the two-tensor transform and the masked epilogue");
        if (a.Cin == 128) return launch_conv_pc2_form<128, 6, true, true, 4, false, false, true>(a, st, "conv3x3 (two boards per unit, 128 channels, gated two-tensor, masked)");
        return launch_conv_pc2_form<256, 5, true, true, 4, false, false, true>(a, st, "conv3x3 (two boards per unit, gated two-tensor, masked)");
    }
    if (a.Cin == 128) {                                        // the 128-channel tower (BASELINE configs[1], keisei-ddp.toml): all 81 squares, no corner launch
        if (a.in2 && a.ep_y) return launch_conv_pc2_form<128, 6, true, true, 4>(a, st, "conv3x3 (two boards per unit, 128 channels, two-tensor, masked)");
        if (a.in2) return launch_conv_pc2_form<128, 6, true, false, 4>(a, st, "conv3x3 (two boards per unit, 128 channels, two-tensor)");
        return launch_conv_pc2_form<128, 6, false, false, 4>(a, st, "conv3x3 (two boards per unit, 128 channels)");
    }
    // border tiles (KA_CONV_PC2_SKIP=1): rows dealt so that three tiles are all halo for three taps each -- 10 % fewer MFMAs and
    // fragment reads, bit-compatible, and SLOWER as built (316 vs 288 us forward, profiles/NOTES_r04.md): ten lane offsets instead of
    // five push the MFMA loop past its 168 registers, and the spill reloads sit on the weight ring's vector-memory counter
    const bool skip = ka_opt(KA_OPT_CONV_PC2_SKIP, 0) != 0;
    const bool stag = ka_opt(KA_OPT_CONV_PC2_STAG, 0) != 0;    // waves 4-7 half a unit behind waves 0-3 (measured level)
    if (stag && !skip) {
        if (a.in2 && a.ep_y) return launch_conv_pc2_form<256, 5, true, true, 4, false, true>(a, st, "conv3x3 (two boards per unit, two-tensor, masked, staggered)");
        if (a.in2) return launch_conv_pc2_form<256, 5, true, false, 4, false, true>(a, st, "conv3x3 (two boards per unit, two-tensor, staggered)");
        return launch_conv_pc2_form<256, 5, false, false, 4, false, true>(a, st, "conv3x3 (two boards per unit, staggered)");
    }
    if (a.in2 && a.ep_y) return launch_conv_pc2_form<256, 5, true, true, 4>(a, st, "conv3x3 (two boards per unit, two-tensor, masked)");
    if (a.in2) return skip ? launch_conv_pc2_form<256, 5, true, false, 4, true>(a, st, "conv3x3 (two boards per unit, two-tensor, border tiles)")
                           : launch_conv_pc2_form<256, 5, true, false, 4>(a, st, "conv3x3 (two boards per unit, two-tensor)");
    return skip ? launch_conv_pc2_form<256, 5, false, false, 4, true>(a, st, "conv3x3 (two boards per unit, border tiles)")
                : launch_conv_pc2_form<256, 5, false, false, 4>(a, st, "conv3x3 (two boards per unit)");
}

// ---------------------------------------------------------------- square 80 of sixteen boards as one row tile
// 81 squares are five row tiles and one square.  As a sixth tile that square cost every (tap, k-step) of every wave two MFMAs
// on 15 zero rows -- 16.7 % of the matrix work and of the activation-fragment reads for 1.2 % of the output (measured: the
// tower kernels are 6-7 % faster without it, tools/_diag/job_r3_j.sh).  The tower kernels therefore compute squares 0..79
// (MT = 5) and this kernel the corner: sixteen boards' corner squares are the 16 rows of ONE tile,
// over K = 4 taps x 256 channels -- only the taps (-1,-1), (-1,0), (0,-1), (0,0) of square (8,8) lie on the board; the
// other five multiplied zeros.  Same weight pack, same input transforms, same (chunk, tap, k-step) order: out[b, 80, :] is
// bit-identical to the six-tile kernels'.  The per-board sums the tower kernel wrote (over 80 squares) receive the corner's
// terms here (+=: one lane per (board, channel), stream-ordered behind the tower kernel; a fixed order of additions).
// The kernel is a chain of latencies (input squares from HBM, weight fragments from the L2, the sums it adds to), so each
// is taken once: a 512-thread workgroup owns 32 boards (two row tiles) x 128 output channels -- 256 workgroups at B = 4096,
// each pulling a 256 KB half of the four taps' weights out of the L2; every thread's eight input pieces are in flight together,
// a wave's 64 weight fragments arrive 12 steps ahead, and the sums to be updated are requested before the MFMA loop.
constexpr int kCornerBoards = 32, kCornerStride = 4 * 512 + 16;      // 2064 B per board: 16 fragment lanes on 16 bank slots
constexpr int kCornerLds = kCornerBoards * kCornerStride;

__global__ __launch_bounds__(512) void conv3x3_corner_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * kCornerBoards;
    const int rt = wave & 1, nt0 = blockIdx.y * 8 + (wave >> 1) * 2;       // this wave: row tile rt, channel tiles nt0, nt0 + 1
    const char* wl = static_cast<const char*>(a.wpack) + (size_t)nt0 * 1024 + lane * 16;
    auto wfrag = [&](int step, int j) {                      // step = (chunk kc, tap slot ps, k-step k4): the tower kernels' order
        const int kc = step >> 4, ps = (step >> 2) & 3, ks = kc * 4 + (step & 3), tap = ps < 2 ? ps : ps + 1;      // taps 0, 1, 3, 4
        return *reinterpret_cast<const bf16x8*>(wl + (size_t)((tap * 8 + ks) * 16 + j) * 1024);
    };
    constexpr int kAhead = 12;                               // (16 ahead spilled at the 256 registers of two waves per SIMD)
    bf16x8 wq[kAhead][2];
#pragma unroll
    for (int s2 = 0; s2 < kAhead; ++s2) { wq[s2][0] = wfrag(s2, 0); wq[s2][1] = wfrag(s2, 1); }
    // ---- stage [board][tap slot][256 channels]: squares 70, 71, 79, 80 = taps 0, 1, 3, 4 of square 80, transformed as the tower
    // kernels' staging transforms them (statement for statement)
    const bool has_aff = a.in_scale != nullptr;
    const int pc = tid & 31, ch0 = pc * 8;                   // this thread's channel piece is the same for all its pieces
    {
        bf16x8 pv[8], pw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                        // 32 boards x 4 squares x 32 pieces = 8 per thread, all in flight
            const int i = tid + 512 * k, bl = i >> 7, ps = (i >> 5) & 3, bb = min(b0 + bl, a.B - 1);
            const int p = ps == 0 ? 70 : ps == 1 ? 71 : ps == 2 ? 79 : 80;
            const size_t off = (((size_t)bb * KA_BOARD + p) * 256 + ch0) * 2;
            pv[k] = *reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in) + off);
            pw[k] = a.in2 ? *reinterpret_cast<const bf16x8*>(static_cast<const char*>(a.in2) + off) : bf16x8{};
        }
        float sc[8], sh[8], k3[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; k3[e] = 0.f; }
        if (has_aff) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.in_scale + ch0), s1 = *reinterpret_cast<const f32x4*>(a.in_scale + ch0 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.in_shift + ch0), t1 = *reinterpret_cast<const f32x4*>(a.in_shift + ch0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = t0[e]; sh[4 + e] = t1[e]; }
        }
        if (a.in2) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.in_k3 + ch0), u1 = *reinterpret_cast<const f32x4*>(a.in_k3 + ch0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { k3[e] = u0[e]; k3[4 + e] = u1[e]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = tid + 512 * k, bl = i >> 7, ps = (i >> 5) & 3;
            bf16x8 v = pv[k];
            if (a.in2 && a.in_bias) {
                // (the tower kernel's gated transform, term for term)
                const size_t go = (size_t)min(b0 + bl, a.B - 1) * 256 + ch0;
                const float* addp = a.in_bias + (size_t)a.B * 256;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.in_bias + go), g1 = *reinterpret_cast<const f32x4*>(a.in_bias + go + 4);
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(addp + go), a1 = *reinterpret_cast<const f32x4*>(addp + go + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ge = e < 4 ? g0[e & 3] : g1[e & 3], ae = e < 4 ? a0[e & 3] : a1[e & 3];
                    v[e] = (__bf16)fmaf((float)pw[k][e], k3[e], fmaf((float)v[e], ge * sc[e], fmaf(ae, sc[e], sh[e])));
                }
            } else if (a.in2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaf((float)pw[k][e], k3[e], fmaf((float)v[e], sc[e], sh[e]));
            } else if (has_aff || a.relu || a.in_bias) {
                f32x4 pb0 = {0.f, 0.f, 0.f, 0.f}, pb1 = pb0;
                if (a.in_bias) {
                    const size_t bo = (size_t)min(b0 + bl, a.B - 1) * 256 + ch0;
                    pb0 = *reinterpret_cast<const f32x4*>(a.in_bias + bo); pb1 = *reinterpret_cast<const f32x4*>(a.in_bias + bo + 4);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[e];
                    if (has_aff) f = fmaf(f, sc[e], sh[e]);
                    if (a.relu) f = fmaxf(f, 0.f);
                    if (a.in_bias) f += e < 4 ? pb0[e & 3] : pb1[e & 3];
                    v[e] = (__bf16)f;
                }
            }
            if (b0 + bl >= a.B) v = bf16x8{};
            *reinterpret_cast<bf16x8*>(cs + bl * kCornerStride + ps * 512 + pc * 16) = v;
        }
    }
    __syncthreads();
    // ---- what the epilogue adds to / compares with is requested now and arrives under the MFMA loop
    const int bb = b0 + 16 * rt + r;                         // lane (r, q): this board, channels cb[j] .. cb[j]+3 of tile j
    const bool live = bb < a.B;
    const int bbc = min(bb, a.B - 1);
    int cb[2];
    f32x4 pbs[2], psq[2], ps1[2], ps2[2], esc[2], esh[2], emu[2], eis[2];
    bf16x4 yv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        cb[j] = chan_of(nt0 + j, 4 * q, 16);
        const size_t srow = (size_t)bbc * 256 + cb[j];
        pbs[j] = psq[j] = ps1[j] = ps2[j] = esc[j] = esh[j] = emu[j] = eis[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        yv[j] = bf16x4{};
        if (a.bsum) pbs[j] = *reinterpret_cast<const f32x4*>(a.bsum + srow);
        if (a.sqpart) psq[j] = *reinterpret_cast<const f32x4*>(a.sqpart + srow);
        if (a.ep_y) {
            ps1[j] = *reinterpret_cast<const f32x4*>(a.ep_s1 + srow); ps2[j] = *reinterpret_cast<const f32x4*>(a.ep_s2 + srow);
            esc[j] = *reinterpret_cast<const f32x4*>(a.ep_scale + cb[j]); esh[j] = *reinterpret_cast<const f32x4*>(a.ep_shift + cb[j]);
            emu[j] = *reinterpret_cast<const f32x4*>(a.ep_mean + cb[j]); eis[j] = *reinterpret_cast<const f32x4*>(a.ep_invstd + cb[j]);
            yv[j] = *reinterpret_cast<const bf16x4*>(static_cast<const char*>(a.ep_y) + (((size_t)bbc * KA_BOARD + 80) * 256 + cb[j]) * 2);
        }
    }
    // ---- 32 k-steps: one activation fragment (this wave's 16 boards), two weight fragments 12 steps ahead, two MFMAs
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* arow = cs + (16 * rt + r) * kCornerStride + q * 16;
#pragma unroll
    for (int step = 0; step < 32; ++step) {
        const bf16x8 w0 = wq[step % kAhead][0], w1 = wq[step % kAhead][1];
        if (step + kAhead < 32) { wq[step % kAhead][0] = wfrag(step + kAhead, 0); wq[step % kAhead][1] = wfrag(step + kAhead, 1); }
        const int ps = (step >> 2) & 3, ks = (step >> 4) * 4 + (step & 3);
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + ps * 512 + ks * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af, acc[1], 0, 0, 0);
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x4 v = acc[j];
        const size_t orow = ((size_t)bb * KA_BOARD + 80) * 256 + cb[j], srow = (size_t)bb * 256 + cb[j];
        if (a.bsum) *reinterpret_cast<f32x4*>(a.bsum + srow) = pbs[j] + v;
        if (a.sqpart) *reinterpret_cast<f32x4*>(a.sqpart + srow) = psq[j] + f32x4{v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]};
        bf16x4 o;
        if (!a.ep_y) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        } else {
            // da = dh * [bn(y) > 0] and its BatchNorm-backward terms (conv_epilogue's masked branch, term for term)
            f32x4 t1, t2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (float)yv[j][e];
                const __bf16 db = (__bf16)v[e];
                const float d = (y * esc[j][e] + esh[j][e] > 0.f) ? (float)db : 0.f;
                t1[e] = d; t2[e] = d * ((y - emu[j][e]) * eis[j][e]);
                o[e] = (__bf16)d;
            }
            *reinterpret_cast<f32x4*>(a.ep_s1 + srow) = ps1[j] + t1;
            *reinterpret_cast<f32x4*>(a.ep_s2 + srow) = ps2[j] + t2;
        }
        *reinterpret_cast<bf16x4*>(static_cast<char*>(a.out) + orow * 2) = o;
    }
}
static int launch_conv_corner(const ConvArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (int rc = ka_big_lds_once(reinterpret_cast<const void*>(&conv3x3_corner_kernel), done, "conv3x3 (corner)")) return rc;
    hipLaunchKernelGGL(conv3x3_corner_kernel, dim3((a.B + kCornerBoards - 1) / kCornerBoards, 2), dim3(512), kCornerLds, st, a);
    return ka_check_launch("conv3x3 (corner)");
}

template <typename T>
int conv_dispatch(ConvArgs a, hipStream_t st) {
    typedef Elem<T> E;
    constexpr int CPK = 4 * E::kPer16;
    KA_REQUIRE(a.B > 0 && a.Cin % CPK == 0 && a.Cout % 16 == 0,
               "conv3x3: need Cin %% %d == 0 and Cout %% 16 == 0 (got Cin=%d Cout=%d)", CPK, a.Cin, a.Cout);
    bool want5 = false;
    if constexpr (sizeof(T) == 2) {
        // the producer / consumer form.  KA_CONV_P: 0 off; 1 (default) the forward forms -- 8 % / 4 % faster alone, 1 % in
        // the step; 2: + the two-tensor data-gradient form with the plain epilogue (no faster alone, slower in the step: it
        // owns every register file, so the weight-gradient stream no longer runs beside it); 3: + the masked epilogue (spills)
        const int pv = ka_opt(KA_OPT_CONV_P, 1);
        // training batches of the 256-channel tower: squares 0..79 as five row tiles here, square 80 of sixteen boards at a
        // time in conv3x3_corner_kernel (KA_CONV_MT=6: all 81 squares as six row tiles, the round-1/2 form)
        want5 = a.Cin == 256 && a.Cout == 256 && a.B >= 512 && ka_opt(KA_OPT_CONV_MT, 5) != 6;
        // two boards per weight fragment (conv3x3_pc2_kernel).  KA_CONV_PC2: 0 off; 1 the forward forms; 2 (default) + the two-tensor
        // data gradient with the register-only epilogue; 3 (default) + the masked epilogue
        const int p2 = ka_opt(KA_OPT_CONV_PC2, 3);
        if (p2 != 0 && want5 && pv != 0 && (!a.in2 || (p2 >= 2 && !a.ep_y) || p2 >= 3)) {
            // square 80: as one more row tile inside the kernel (mt5 = 2, KA_CONV_CORNER_IN) or by conv3x3_corner_kernel behind it
            // (KA_CONV_CORNER_IN: 1, default = inside the forward forms and the plain-epilogue data gradient; 0 = the launch everywhere.
            //  The masked form keeps the launch: inside, its tile epilogue -- eight sums read back -- measured 5-7 us slower)
            const bool corner_in = ka_opt(KA_OPT_CONV_CORNER_IN, 1) != 0 && !a.ep_y && ka_opt(KA_OPT_CONV_PC2_SKIP, 0) == 0 &&
                                   ka_opt(KA_OPT_CONV_PC2_STAG, 0) == 0;
            a.mt5 = corner_in ? 2 : 1;
            if (int rc = launch_conv_pc2(a, st)) return rc;
            return corner_in ? KA_OK : launch_conv_corner(a, st);
        }
        if (p2 != 0 && pv != 0 && a.Cin == 128 && a.Cout == 128 && a.B >= 512 && (!a.in2 || (p2 >= 2 && !a.ep_y) || p2 >= 3))
            return launch_conv_pc2(a, st);
        KA_REQUIRE(!(a.in2 && a.in_bias), "conv3x3: the gated two-tensor input exists in the two-board kernel only (ka_conv3x3_dgrad_gated_supported)");
        if (pv != 0 && a.Cin == 256 && a.Cout == 256 && a.B >= 512 &&
            (!a.in2 || (pv >= 2 && !a.ep_y) || pv >= 3)) {
            a.mt5 = want5;
            if (int rc = launch_conv_pc(a, st)) return rc;
            return want5 ? launch_conv_corner(a, st) : KA_OK;
        }
    }
    // boards per workgroup: 1 = 256-thread workgroups, two independent ones per CU when the tile allows it
    int wm = 1;
    if (const int v = ka_opt(KA_OPT_CONV_WM, 0); v == 1 || v == 2) wm = v;   // experiments
    const int img_squares = wm == 2 ? kLdsSquares : kImgSquares1;
    // LDS chunk: the largest divisor of Cin (in k-steps) whose 16-byte pieces tile the 256 staging threads of a board
    // and whose image fits: squares x (KC*size + 32) <= 150 KiB
    int kc = 0;
    for (int steps = a.Cin / CPK; steps >= 1; --steps) {
        const int cand = steps * CPK;
        if (a.Cin % cand != 0 || 256 % (cand * E::kSize / 16) != 0) continue;
        if ((size_t)(img_squares + kZeroSquares) * (cand * E::kSize + 32) > 150 * 1024) continue;
        kc = cand;
        break;
    }
    KA_REQUIRE(kc > 0, "conv3x3: cannot chunk Cin=%d", a.Cin);
    // measured (bf16, C=256): 2 chunks of 128 beat one of 256 -- with one board per workgroup the smaller tile lets two
    // workgroups share a CU, and inside the training step that is worth 15 %
    if (E::kSize == 2 && kc > 128 && a.Cin % 128 == 0) kc = 128;
    int ntw = a.Cout > 128 ? 4 : (a.Cout > 64 ? 2 : 1);
    // small batches (rollout inference): narrower channel slabs until there is about one workgroup per CU -- measured
    // at C = 256: 128 boards 34 -> 27 us with 2 tiles per wave, 64 boards 34 -> 24 us with 1
    while (ntw > 1 && (long long)((a.B + wm - 1) / wm) * ((a.Cout + 64 * ntw - 1) / (64 * ntw)) < 256) ntw >>= 1;
    a.tune_prio = 1;          // static priority for the second wave of every SIMD: it reaches its epilogue first
    // tuning overrides (experiments only): channels per LDS chunk, n-tiles per wave
    if (const int v = ka_opt(KA_OPT_CONV_KC, 0); v > 0 && a.Cin % v == 0 && v % CPK == 0 && v <= kc) kc = v;
    if (const int v = ka_opt(KA_OPT_CONV_NTW, 0); v == 1 || v == 2 || v == 4) ntw = v;
    a.tune_stagger = ka_opt(KA_OPT_CONV_STAGGER, a.tune_stagger);
    a.tune_prio = ka_opt(KA_OPT_CONV_PRIO, a.tune_prio);
    KA_REQUIRE(256 % (kc * E::kSize / 16) == 0, "conv3x3: chunk of %d channels does not tile the workgroup", kc);
    a.KC = kc;
    if (wm == 1) {
        if (ntw == 4) {
            if constexpr (sizeof(T) == 2) {
                if (want5) {
                    a.mt5 = 1;
                    if (int rc = launch_conv<T, 4, 1>(a, st)) return rc;
                    return launch_conv_corner(a, st);
                }
            }
            return launch_conv<T, 4, 1>(a, st);
        }
        if (ntw == 2) return launch_conv<T, 2, 1>(a, st);
        return launch_conv<T, 1, 1>(a, st);
    }
    if (ntw == 4) return launch_conv<T, 4, 2>(a, st);
    if (ntw == 2) return launch_conv<T, 2, 2>(a, st);
    return launch_conv<T, 1, 2>(a, st);
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" int ka_conv3x3_fwd(const void* in, const void* wpack, void* out, const float* in_scale,
                              const float* in_shift, const float* in_bias, int relu, float* bsum, float* sqpart,
                              int B, int Cin, int Cout, int dtype, void* stream) {
    ConvArgs a{in, wpack, out, in_scale, in_shift, in_bias, bsum, sqpart, B, Cin, Cout, 0, relu,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, g_stamps.load()};
    KA_REQUIRE(in && wpack && out, "conv3x3: null tensor");
    KA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3: scale/shift must come together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KA_DTYPE_BF16) return conv_dispatch<bf16_t>(a, st);
    if (dtype == KA_DTYPE_F32) return conv_dispatch<float>(a, st);
    ka_set_error("conv3x3: unknown dtype %d", dtype);
    return KA_ERR_ARG;
}

// ka_conv3x3_fwd that also WRITES the transformed input x' = [relu](in * in_scale + in_shift) + in_bias -- the tensor the convolution
// actually multiplies -- to x_out (B, 81, Cin): the weight gradient of the layer then reads it as a plain operand instead of
// repeating the transform per tile (wgrad_flat_kernel<true>: +37 us per launch at B = 4096).  Only the two-board kernel's staging
// waves can do it for nothing (ka_conv3x3_fwd_keep_supported); x_out == NULL is ka_conv3x3_fwd.
extern "C" int ka_conv3x3_fwd_keep_supported(int B, int Cin, int Cout, int dtype) {
    if (dtype != KA_DTYPE_BF16 || Cin != Cout || B < 512) return 0;
    if (ka_opt(KA_OPT_CONV_P, 1) == 0 || ka_opt(KA_OPT_CONV_PC2, 3) == 0) return 0;
    return Cin == 256 && ka_opt(KA_OPT_CONV_MT, 5) != 6;
}
extern "C" int ka_conv3x3_fwd_keep(const void* in, const void* wpack, void* out, const float* in_scale, const float* in_shift,
                                   const float* in_bias, int relu, float* bsum, float* sqpart, void* x_out, int B, int Cin, int Cout,
                                   int dtype, void* stream) {
    KA_REQUIRE(in && wpack && out, "conv3x3: null tensor");
    KA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3: scale/shift must come together");
    KA_REQUIRE(!x_out || ((in_scale || in_bias || relu) && ka_conv3x3_fwd_keep_supported(B, Cin, Cout, dtype)),
               "conv3x3_fwd_keep: x_out needs an input transform and a shape the two-board kernel takes (B=%d Cin=%d Cout=%d)", B, Cin, Cout);
    ConvArgs a{in, wpack, out, in_scale, in_shift, in_bias, bsum, sqpart, B, Cin, Cout, 0, relu,
               nullptr, nullptr, x_out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, g_stamps.load()};
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
               in2, k + 2 * Cin, dy_out, ep_y, ep_scale, ep_shift, ep_mean, ep_invstd, ep_s1, ep_s2, 0, 0, g_stamps.load()};
    return conv_dispatch<bf16_t>(a, static_cast<hipStream_t>(stream));
}

// does conv_dispatch take a two-tensor launch of this shape to the two-board kernel (the only form with the gated input)?
extern "C" int ka_conv3x3_dgrad_gated_supported(int B, int Cin, int Cout, int dtype, int masked) {
    if (dtype != KA_DTYPE_BF16 || B < 512 || Cin != Cout || (Cin != 256 && Cin != 128)) return 0;
    const int pv = ka_opt(KA_OPT_CONV_P, 1), p2 = ka_opt(KA_OPT_CONV_PC2, 3);
    if (pv == 0 || p2 < 3 || !masked) return 0;
    if (Cin == 256 && ka_opt(KA_OPT_CONV_MT, 5) == 6) return 0;
    return 1;
}

// ka_conv3x3_dgrad_fused whose gradient input is given as du and gate_add = [gate | add] ([2][B][Cin] fp32) with
// dz = du * gate[b, c] + add[b, c] (ka_block_dx_tail_bwd_du_gate writes the three): dy = dz*k[0] + k[1] + in2*k[2] is formed from the unrounded dz, written
// to dy_out and convolved.  Everything else as ka_conv3x3_dgrad_fused.
extern "C" int ka_conv3x3_dgrad_fused_gated(const void* du, const float* gate_add, const void* in2, const float* k,
                                            void* dy_out, const void* wpack, void* out, float* bsum, const void* ep_y,
                                            const float* ep_scale, const float* ep_shift, const float* ep_mean,
                                            const float* ep_invstd, float* ep_s1, float* ep_s2, int B, int Cin, int Cout,
                                            int dtype, void* stream) {
    KA_REQUIRE(du && gate_add && in2 && k && wpack && out, "conv3x3_dgrad_fused_gated: null tensor");
    KA_REQUIRE(dtype == KA_DTYPE_BF16, "conv3x3_dgrad_fused_gated: bf16 only");
    KA_REQUIRE(!ep_y || (ep_scale && ep_shift && ep_mean && ep_invstd && ep_s1 && ep_s2), "conv3x3_dgrad_fused_gated: epilogue tensors");
    KA_REQUIRE(ka_conv3x3_dgrad_gated_supported(B, Cin, Cout, dtype, ep_y != nullptr),
               "conv3x3_dgrad_fused_gated: shape B=%d C=%d/%d %s is not taken by the two-board kernel's masked form", B, Cin, Cout,
               ep_y ? "masked" : "plain");
    ConvArgs a{du, wpack, out, k, k + Cin, gate_add, bsum, nullptr, B, Cin, Cout, 0, 0,
               in2, k + 2 * Cin, dy_out, ep_y, ep_scale, ep_shift, ep_mean, ep_invstd, ep_s1, ep_s2, 0, 0, g_stamps.load()};
    return conv_dispatch<bf16_t>(a, static_cast<hipStream_t>(stream));
}

// diagnostic: stamps != null makes every conv3x3 workgroup record 4 s_memtime values and 2 s_memrealtime values (100 MHz ticks are NOT used:
// s_memtime counts shader clocks); pass null to switch off
extern "C" int ka_debug_conv_stamps(unsigned long long* stamps) { g_stamps.store(stamps); return KA_OK; }

extern "C" int ka_conv3x3_sqpart_rows(int B) { return B; }

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

// table: device int64 [n][8] = {src weight pointer, dst pack pointer, Co, Ci, Nout, Kin, mode, 0} (see ka_pack_conv3x3
// for the meaning of each); max_pieces = the largest 9*(Kin/cpk)*(Nout/16)*64 among the entries (sizes the grid)
extern "C" int ka_pack_conv3x3_multi(const long long* table, int n, long long max_pieces, int dtype, void* stream) {
    KA_REQUIRE(table && n > 0 && max_pieces > 0, "pack_conv3x3_multi: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bx = (int)((max_pieces + 255) / 256 < 64 ? (max_pieces + 255) / 256 : 64);
    if (dtype == KA_DTYPE_BF16)
        hipLaunchKernelGGL(pack_conv3x3_multi_kernel<bf16_t>, dim3(bx, n), dim3(256), 0, st, table);
    else if (dtype == KA_DTYPE_F32)
        hipLaunchKernelGGL(pack_conv3x3_multi_kernel<float>, dim3(bx, n), dim3(256), 0, st, table);
    else { ka_set_error("pack_conv3x3_multi: unknown dtype %d", dtype); return KA_ERR_ARG; }
    return ka_check_launch("pack_conv3x3_multi");
}
