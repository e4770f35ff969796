// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate per SIMD with 1, 2, 3 waves per SIMD, independent accumulators (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_rate(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = (float)lane, b = (float)(lane * 3);

    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (s == 1.2345f) sink[0] = s;
}

template <int NACC>
static void run(int waves) {
    unsigned long long* d; float* sink; (void)hipMalloc(&d, 256 * 16 * 8); (void)hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mfma_rate<NACC>, dim3(256), dim3(64 * waves), 0, 0, d, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (int w = 0; w < waves; ++w) cyc += (double)h[16 + w]; cyc /= waves;
    const double per_simd = (double)iters * NACC * (waves / 4.0);
    printf("%2d independent accumulators, %2d waves/CU (%d per SIMD): %.1f cycles per MFMA per wave, %.1f cycles per MFMA per SIMD\n", NACC, waves, waves / 4,
           cyc / (iters * (double)NACC), cyc / per_simd);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    for (int w : {4, 8, 12}) { run<4>(w); run<16>(w);  }
    return 0;
}
