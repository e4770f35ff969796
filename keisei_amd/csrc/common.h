// Shared device helpers for the keisei_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KA_OK 0
#define KA_ERR_ARG (-1)
#define KA_ERR_HIP (-2)
#define KA_ERR_UNSUPPORTED (-3)

#define KA_DTYPE_F32 0
#define KA_DTYPE_BF16 1

#define KA_BOARD 81       // 9x9 squares
#define KA_PADBOARD 121   // 11x11 zero-haloed board

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct bf16_t { uint16_t v; };   // storage type for bf16 tensors

// ---- error plumbing (host) -------------------------------------------------
void ka_set_error(const char* fmt, ...);
int ka_check_launch(const char* what);
#define KA_REQUIRE(cond, ...) do { if (!(cond)) { ka_set_error(__VA_ARGS__); return KA_ERR_ARG; } } while (0)

// Kernels that need more than 64 KiB of dynamic LDS: hipFuncAttributeMaxDynamicSharedMemorySize is a per-device
// property of the function, so it is set once per (kernel instantiation, device) -- `done` is that instantiation's
// bit mask of devices already configured (the only state the library keeps; written once per device, thread-safe).
#include <atomic>
inline int ka_big_lds_once(const void* func, std::atomic<unsigned long long>& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { ka_set_error("%s: hipGetDevice failed", what); return KA_ERR_HIP; }
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return KA_OK;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        ka_set_error("%s: hipFuncSetAttribute failed on device %d", what, dev);
        return KA_ERR_HIP;
    }
    done.fetch_or(bit, std::memory_order_release);
    return KA_OK;
}

// Ablation switches (KA_CONV_P_ABL, KA_CONV_T_ABL, KA_TOWER_ABL: they skip phases of a kernel and give WRONG results)
// exist only in diagnostic builds -- `KA_DIAG=1 python -m keisei_amd.build` compiles libkeisei_amd_diag.so with -DKA_DIAG;
// the product library never reads them.  (Switches that choose between two CORRECT implementations stay run-time; bench.py
// records every KA_* variable set in its environment.)
#include <stdlib.h>
inline const char* ka_diag_env(const char* name) {
#ifdef KA_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- run-time switches (launch plan) ----------------------------------------------------------------------------------------
// Every KA_* switch the library honours is read from the environment ONCE -- at the first launch that asks, or again when
// the caller says so (ka_options_reload: tests and A/B tools that flip a switch inside one process) -- into a table the
// launchers index: no getenv() on the enqueue path.  bench.py records the switches set in its environment.
#define KA_OPTIONS(X)                                                                                                      \
    X(CONV_P) X(CONV_MT) X(CONV_PC2) X(CONV_PC2_SKIP) X(CONV_PC2_STAG) X(CONV_CORNER_IN) X(CONV_P_STAG) X(CONV_P_PRIO) X(CONV_P_WGS) X(CONV_P_NPW) X(CONV_WM) X(CONV_KC)       \
    X(CONV_NTW) X(CONV_STAGGER) X(CONV_PRIO) X(WGRAD_TN) X(WGRAD_WGS) X(WGRAD_STAG) X(WGRAD_LEAN) X(BOARD_PAIRS) X(TF_LDS_EPI) X(TF_K256) X(TF_BIG)    \
    X(TF_MAP2D) X(TF_ATTN_LDS) X(TF_ATTN_ONE) X(TAIL_FWD_KB) X(TAIL_GATE_P4)
enum KaOpt {
#define KA_OPT_ENUM(n) KA_OPT_##n,
    KA_OPTIONS(KA_OPT_ENUM)
#undef KA_OPT_ENUM
    KA_OPT_COUNT
};
struct KaOptVal { int set, val; };
const KaOptVal* ka_opts();                                   // capi.hip
inline bool ka_opt_set(KaOpt o) { return ka_opts()[o].set != 0; }
inline int ka_opt(KaOpt o, int dflt) { const KaOptVal v = ka_opts()[o]; return v.set ? v.val : dflt; }

// diagnostic only: [workgroup][8] stamp buffer of ka_debug_conv_stamps (slots 0 / 7: s_memtime at the start / end of the workgroup's
// main loop, 3 / 4: s_memrealtime there); null in production -- no stamp executes
std::atomic<unsigned long long*>& ka_debug_stamps();         // capi.hip

// ---- scalar conversions -----------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// Plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950.
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kSize = 4;
    static constexpr int kPer16 = 4;      // elements per 16-byte piece
    static constexpr int kDtype = KA_DTYPE_F32;
    typedef f32x4 vec16;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void unpack(const vec16& v, float* f) {
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
    static __device__ __forceinline__ vec16 pack(const float* f) { vec16 v = {f[0], f[1], f[2], f[3]}; return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kSize = 2;
    static constexpr int kPer16 = 8;
    static constexpr int kDtype = KA_DTYPE_BF16;
    typedef bf16x8 vec16;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(p->v); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f2bf(v); }
    static __device__ __forceinline__ void unpack(const vec16& v, float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ vec16 pack(const float* f) {
        vec16 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
        return v;
    }
};

// pair (2-element) access used by the board kernels: one thread owns 2 adjacent channels
template <typename T> __device__ __forceinline__ f32x2 ld2(const T* p);
template <> __device__ __forceinline__ f32x2 ld2<float>(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
template <> __device__ __forceinline__ f32x2 ld2<bf16_t>(const bf16_t* p) {
    uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    f32x2 r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
    return r;
}
template <typename T> __device__ __forceinline__ void st2(T* p, f32x2 v);
template <> __device__ __forceinline__ void st2<float>(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t* p, f32x2 v) {
    uint32_t u = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    *reinterpret_cast<uint32_t*>(p) = u;
}
// round-trip through the storage type (so statistics see exactly what is stored)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }

// padded-board index of square p (0..80) on the 11x11 haloed grid
__device__ __forceinline__ int pad_index(int p) { return (p / 9 + 1) * 11 + (p % 9) + 1; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it waits
// for every global load in flight (a weight prefetch issued before the barrier) and for global stores to complete
#define KA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
