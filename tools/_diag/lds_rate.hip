// Micro-benchmark: LDS read throughput per CU for the fragment reads of the MFMA kernels (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/_diag/lds_rate.hip -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef bf16x4 __attribute__((address_space(3)))* lds_bf16x4_ptr;

template <int MODE>   // 0: ds_read_b64_tr_b16 (rows 288 B apart, the wgrad A pattern), 1: ds_read_b128 (lane-contiguous), 2: ds_read_b64 plain, 3: tr, rows 160 B apart (X squares)
__global__ __launch_bounds__(1024) void lds_rate(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    for (int i = tid; i < 40000; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    int base;
    if (MODE == 0) base = (4 * q + (r >> 2)) * 288 + 8 * (r & 3) + (wave & 3) * 32;
    else if (MODE == 3) base = (4 * q + (r >> 2)) * 160 + 8 * (r & 3) + (wave & 3) * 32;
    else if (MODE == 1) base = lane * 16 + wave * 1024;
    else base = lane * 8 + wave * 512;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = base + u * (MODE == 3 ? 16 * 160 : 16 * 288) % 60000;
            if (MODE == 0 || MODE == 3) {
                bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + off));
                acc ^= __builtin_bit_cast(u32x2, v)[0];
            } else if (MODE == 1) {
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + (base + u * 16384) % 131072);
                acc ^= v[0] ^ v[3];
            } else {
                u32x2 v = *reinterpret_cast<const u32x2*>(smem + (base + u * 8192) % 131072);
                acc ^= v[0];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char* name, int waves, int bytes_per_instr) {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_rate<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(lds_rate<MODE>, dim3(256), dim3(64 * waves), 160 * 1024, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (int w = 0; w < waves; ++w) cyc += (double)h[16 + w]; cyc /= waves;     // workgroup 1
    const double bytes = (double)waves * iters * 16 * bytes_per_instr;
    printf("%-44s %2d waves/CU: %7.1f B/clk/CU  (%.1f clk per wave-instruction)\n", name, waves, bytes / cyc, cyc / (iters * 16.0 * waves));
    hipFree(d);
}

int main() {
    for (int w : {4, 8, 16}) {
        run<0>("ds_read_b64_tr_b16, rows 288 B apart", w, 512);
        run<3>("ds_read_b64_tr_b16, rows 160 B apart", w, 512);
        run<2>("ds_read_b64", w, 512);
        run<1>("ds_read_b128", w, 1024);
    }
    return 0;
}
