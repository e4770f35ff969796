// Diagnostic only (not part of the library): an MFMA-bound kernel with a chosen register / LDS footprint, to measure what a
// co-resident HBM-bound board kernel costs it and vice versa (tools/_diag/coresidency.py).
//   512 threads per workgroup, one workgroup per CU, ~150 VGPRs (2 waves per SIMD -> ~300 of 512 registers per SIMD lane),
//   `lds` bytes of dynamic LDS declared (touched once), `iters` x 32 MFMAs per wave on register operands.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" __global__ __launch_bounds__(512, 1) void mfma_burn_kernel(float* out, int iters, int lds_words) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_words; i += 512) smem[i] = (float)i;
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(0.37f * ((tid * 7 + i * 13 + e * 5) % 17) - 3.f);       // non-trivial operands (power!)
            b[i][e] = (__bf16)(0.21f * ((tid * 11 + i * 3 + e * 7) % 19) - 2.f);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = smem[(tid * 31) % (lds_words > 0 ? lds_words : 1)];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
}

// the same FLOPs per iteration as 16 v_mfma_f32_32x32x16_bf16 (half the instructions and operand-register reads per FLOP)
typedef __attribute__((ext_vector_type(16))) float f32x16;
extern "C" __global__ __launch_bounds__(512, 1) void mfma_burn32_kernel(float* out, int iters, int lds_words) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_words; i += 512) smem[i] = (float)i;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(0.37f * ((tid * 7 + i * 13 + e * 5) % 17) - 3.f);
            b[i][e] = (__bf16)(0.21f * ((tid * 11 + i * 3 + e * 7) % 19) - 2.f);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i & 7], 0, 0, 0);
    }
    float s = smem[(tid * 31) % (lds_words > 0 ? lds_words : 1)];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + tid] = s;
}
extern "C" int mfma_burn32(float* out, int wgs, int iters, int lds_bytes, void* stream) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)mfma_burn32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(mfma_burn32_kernel, dim3(wgs), dim3(512), lds_bytes, (hipStream_t)stream, out, iters, lds_bytes / 4);
    return (int)hipGetLastError();
}

extern "C" int mfma_burn(float* out, int wgs, int iters, int lds_bytes, void* stream) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)mfma_burn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(mfma_burn_kernel, dim3(wgs), dim3(512), lds_bytes, (hipStream_t)stream, out, iters, lds_bytes / 4);
    return (int)hipGetLastError();
}
