// Measures the issue rate of v_fma_f32 and v_pk_fma_f32 on gfx950: wave64 instructions per
// cycle per SIMD at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <bool PACKED> __global__ void __launch_bounds__(1024) rate(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f, c = 1e-9f;
    const v2f m2 = {m, m}, c2 = {c, c};
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if constexpr (!PACKED) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                             : "v"(m2), "v"(c2));
            }
        }
    }
    long long t1 = clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    float *out;
    long long *cyc;
    hipMalloc(&out, sizeof(float) * 1024 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 4096);
    const int iters = 20000;
    printf("CUs %d, clock %d kHz\n", n_cu, clk_khz);
    for (int packed = 0; packed < 2; packed++)
        for (int block : {256, 512, 1024}) { // 1, 2, 4 waves per SIMD (one workgroup per CU)
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (packed)
                    hipLaunchKernelGGL(rate<true>, dim3(n_cu), dim3(block), 0, 0, out, cyc, iters);
                else
                    hipLaunchKernelGGL(rate<false>, dim3(n_cu), dim3(block), 0, 0, out, cyc, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(n_cu);
            hipMemcpy(h.data(), cyc, sizeof(long long) * n_cu, hipMemcpyDeviceToHost);
            double instr_per_wave = (double)iters * 64;
            double waves_per_simd = block / 256.0;
            // clock64 ticks at a fixed 100 MHz on this GPU family; use the event time and the shader clock
            double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
            double lanes = (packed ? 2.0 : 1.0) * 64;
            double tflops = 2.0 * lanes * instr_per_wave * waves_per_simd * 4 * n_cu / (ms * 1e-3) / 1e12;
            printf("%s waves/SIMD %.0f: %.3f ms, %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz), %.1f TFLOP/s, clock64 delta %lld\n",
                   packed ? "v_pk_fma_f32" : "v_fma_f32   ", waves_per_simd, ms, ns_per_instr_per_simd,
                   ns_per_instr_per_simd * 2.4, tflops, h[0]);
        }
    return 0;
}
