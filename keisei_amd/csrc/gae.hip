// Batched Generalised Advantage Estimation: one thread per environment walks its trajectory
// backwards; every step is a coalesced access along the environment axis of the (T,N) grids.
// One launch replaces the reference's ~2T launches (gae.py:192-218) and folds in the NaN
// bootstrap-override select and the padded / per-env-length variant (gae.py:261-296).
// Compiled with FP contraction off: the result is bit-identical to the reference's op order
//   delta = (r + (gamma*nv)*nd) - v ;  A_t = delta + ((gamma*lam)*nd) * A_{t+1}.
#include "common.h"
#pragma clang fp contract(off)

namespace {

template <typename F>
__global__ void gae_kernel(const F* __restrict__ rewards, const F* __restrict__ values, const float* __restrict__ term,
                           const F* __restrict__ next_value, const F* __restrict__ override_,
                           const long long* __restrict__ lengths, F* __restrict__ adv, int T, int N, F gamma,
                           float gamma_lam) {
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const F boot = next_value[n];
    long long last = T - 1;
    if (lengths) { last = lengths[n] - 1; if (last < 0) last = 0; }
    F gae = 0;
    F vnext = boot;
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * N + n;
        const F v = values[i];
        F nv = (t == T - 1 || t == last) ? boot : vnext;
        if (override_) { const F o = override_[i]; if (o == o) nv = o; }
        const float nd = 1.0f - term[i];
        const F delta = (rewards[i] + (gamma * nv) * (F)nd) - v;
        const F decay = (F)(gamma_lam * nd);
        gae = delta + decay * gae;
        adv[i] = gae;
        vnext = v;
    }
}

// (x - mean) / (std_unbiased + 1e-8) over a flat vector; single workgroup, two passes (katago_ppo.py:797-798)
__global__ __launch_bounds__(1024) void normalize_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
    __shared__ double red[16];
    __shared__ double bc[2];
    double s = 0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += red[w]; bc[0] = t / (double)n; }
    __syncthreads();
    const double mean = bc[0];
    double q = 0;
    for (long long i = threadIdx.x; i < n; i += 1024) { const double d = x[i] - mean; q += d * d; }
    q = wave_sum_d(q);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += red[w]; bc[1] = sqrt(t / (double)(n - 1)); }
    __syncthreads();
    const float m = (float)mean, sd = (float)bc[1];
    for (long long i = threadIdx.x; i < n; i += 1024) out[i] = (x[i] - m) / (sd + 1e-8f);
}

}  // namespace

// f64 != 0: rewards/values/next_value/override/adv are double; `term` is always float (0/1)
extern "C" int ka_gae(const void* rewards, const void* values, const float* term, const void* next_value,
                      const void* override_, const long long* lengths, void* adv, int T, int N, double gamma,
                      double lam, int f64, void* stream) {
    KA_REQUIRE(rewards && values && term && next_value && adv && T > 0 && N > 0, "gae: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float gl = (float)(gamma * lam);      // python-float product, then fp32 (gae.py:190)
    dim3 grid((N + 63) / 64), block(64);
    if (f64)
        hipLaunchKernelGGL(gae_kernel<double>, grid, block, 0, st, (const double*)rewards, (const double*)values, term,
                           (const double*)next_value, (const double*)override_, lengths, (double*)adv, T, N, gamma, gl);
    else
        hipLaunchKernelGGL(gae_kernel<float>, grid, block, 0, st, (const float*)rewards, (const float*)values, term,
                           (const float*)next_value, (const float*)override_, lengths, (float*)adv, T, N, (float)gamma, gl);
    return ka_check_launch("gae");
}

extern "C" int ka_normalize_advantages(const float* x, float* out, long long n, void* stream) {
    KA_REQUIRE(x && out && n > 1, "normalize_advantages: need n > 1");
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, out, n);
    return ka_check_launch("normalize_advantages");
}
