// Global-norm gradient clipping + Adam over the whole parameter set in three launches
// (multi-tensor): per-chunk sum of squares -> one finalize block (norm, clip coefficient,
// inf/NaN + guard-flag skip decision, GradScaler bookkeeping) -> fused unscale+clip+Adam.
// Everything the reference does with three host round trips per step (GradScaler.unscale_
// found-inf .item(), clip_grad_norm_, optimizer.step, scaler.update) stays on the device.
//
// Reference: keisei/training/katago_ppo.py:926-933 (GradScaler + clip_grad_norm_ + torch Adam,
// lr 2e-4, betas (0.9, 0.999), eps 1e-8, no weight decay).
#include "common.h"

namespace {

constexpr int kChunk = 4096;      // elements per block

struct TensorRef { float* p; const float* g; float* m; float* v; long long n; };

// blk_tensor[b], blk_off[b] map a block to (tensor, element offset)
__global__ __launch_bounds__(256) void sqnorm_kernel(const TensorRef* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                     const long long* __restrict__ blk_off, double* __restrict__ partial) {
    const TensorRef t = tab[blk_tensor[blockIdx.x]];
    const long long o = blk_off[blockIdx.x];
    const long long end = min(t.n, o + (long long)kChunk);
    double s = 0.0;
    for (long long i = o + threadIdx.x; i < end; i += 256) { const float g = t.g[i]; s += (double)g * (double)g; }
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// state[0]=step count (float), ctl[0]=unscaled grad norm, ctl[1]=grad multiplier (clip/scale), ctl[2]=skip flag
// scaler[0]=scale, scaler[1]=growth tracker (as float) -- torch.amp.GradScaler semantics (growth 2, backoff .5, 2000)
__global__ void gnorm_finalize_kernel(const double* __restrict__ partial, int nblocks, float max_norm,
                                      float* __restrict__ scaler, const int* __restrict__ guard_flags,
                                      float* __restrict__ ctl, float* __restrict__ acc_gnorm, float growth_factor,
                                      float backoff_factor, int growth_interval) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
        const float scale = scaler ? scaler[0] : 1.f;
        const float norm = (float)(sqrt(tot) / (double)scale);
        const bool bad = !(norm == norm) || isinf(norm);
        const bool guard = guard_flags && (guard_flags[0] | guard_flags[1]);
        float coef = max_norm / (norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
        ctl[0] = norm;
        ctl[1] = coef / scale;
        ctl[2] = (bad || guard) ? 1.f : 0.f;
        if (acc_gnorm) *acc_gnorm += norm;
        if (scaler) {
            if (bad) { scaler[0] = scale * backoff_factor; scaler[1] = 0.f; }
            else {
                const float tr = scaler[1] + 1.f;
                if ((int)tr >= growth_interval) { scaler[0] = scale * growth_factor; scaler[1] = 0.f; }
                else scaler[1] = tr;
            }
        }
    }
}

// step_state[0]: number of applied steps (incremented by adam_step_kernel block 0 AFTER use via a
// separate tiny kernel to avoid a race) -- here each block reads the pre-incremented value.
__global__ __launch_bounds__(256) void adam_kernel(const TensorRef* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                   const long long* __restrict__ blk_off, const float* __restrict__ ctl,
                                                   const float* __restrict__ step_state, float lr, float beta1,
                                                   float beta2, float eps) {
    if (ctl[2] != 0.f) return;                       // skipped step: weights and moments untouched
    const TensorRef t = tab[blk_tensor[blockIdx.x]];
    const long long o = blk_off[blockIdx.x];
    const long long end = min(t.n, o + (long long)kChunk);
    const float gm = ctl[1];
    const double step = (double)step_state[0] + 1.0;
    const float bc1 = (float)(1.0 - pow((double)beta1, step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, step));
    const float step_size = lr / bc1;
    for (long long i = o + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i] * gm;
        float m = t.m[i], v = t.v[i];
        m = m + (g - m) * (1.f - beta1);
        v = v * beta2 + (1.f - beta2) * g * g;
        t.m[i] = m; t.v[i] = v;
        t.p[i] = t.p[i] - step_size * (m / (sqrtf(v) / bc2s + eps));
    }
}

__global__ void adam_step_inc_kernel(const float* __restrict__ ctl, float* __restrict__ step_state) {
    if (ctl[2] == 0.f) step_state[0] += 1.f;
}

}  // namespace

extern "C" int ka_adam_chunk(void) { return kChunk; }

// tab: device array of nt {p,g,m,v,n} records (5 x 8 bytes each); blk_*: device arrays of nblocks entries
extern "C" int ka_clip_adam_step(const void* tab, const int* blk_tensor, const long long* blk_off, int nblocks,
                                 double* partial, float* ctl, float* step_state, float* scaler,
                                 const int* guard_flags, float* acc_gnorm, float max_norm, float lr, float beta1,
                                 float beta2, float eps, void* stream) {
    KA_REQUIRE(tab && blk_tensor && blk_off && partial && ctl && step_state && nblocks > 0, "clip_adam_step: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const TensorRef* t = static_cast<const TensorRef*>(tab);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nblocks), dim3(256), 0, st, t, blk_tensor, blk_off, partial);
    hipLaunchKernelGGL(gnorm_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblocks, max_norm, scaler,
                       guard_flags, ctl, acc_gnorm, 2.0f, 0.5f, 2000);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, st, t, blk_tensor, blk_off, ctl, step_state, lr, beta1,
                       beta2, eps);
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, st, ctl, step_state);
    return ka_check_launch("clip_adam_step");
}
