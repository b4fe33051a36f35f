// Issue rate of the VALU instructions a ray tracer is made of (gfx950, wave64, 4 waves per SIMD):
// cycles per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S S S S S S S S
#define BODY(INS)                                                                                   \
    for (int i = 0; i < iters; i++) {                                                                \
        asm volatile(REP8(INS " %0, %0, %8\n " INS " %1, %1, %8\n " INS " %2, %2, %8\n " INS " %3, %3, %8\n" \
                          INS " %4, %4, %8\n " INS " %5, %5, %8\n " INS " %6, %6, %8\n " INS " %7, %7, %8\n") \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                     : "v"(m));                                                                      \
    }
#define BODY1(INS)                                                                                  \
    for (int i = 0; i < iters; i++) {                                                                \
        asm volatile(REP8(INS " %0, %0\n " INS " %1, %1\n " INS " %2, %2\n " INS " %3, %3\n"       \
                          INS " %4, %4\n " INS " %5, %5\n " INS " %6, %6\n " INS " %7, %7\n")        \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    }

template <int OP> __global__ void __launch_bounds__(1024) rate(float *out, int iters) {
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    if constexpr (OP == 0) { BODY("v_mul_f32") }
    if constexpr (OP == 1) { BODY("v_min_f32") }
    if constexpr (OP == 2) { BODY("v_add_u32") }
    if constexpr (OP == 3) { BODY("v_and_b32") }
    if constexpr (OP == 4) { BODY("v_cndmask_b32") } // vcc as the mask
    if constexpr (OP == 5) {
        for (int i = 0; i < iters; i++)
            asm volatile(REP8("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                              "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    if constexpr (OP == 6) {
        for (int i = 0; i < iters; i++)
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
    }
    if constexpr (OP == 7) { BODY1("v_rcp_f32") }
    if constexpr (OP == 8) { BODY1("v_sqrt_f32") }
    if constexpr (OP == 9) { BODY1("v_sin_f32") }
    if constexpr (OP == 10) { BODY1("v_mov_b32") }
    if constexpr (OP == 11) {
        for (int i = 0; i < iters; i++)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    if constexpr (OP == 12) { // v_cndmask with an SGPR-pair mask (VOP3)
        for (int i = 0; i < iters; i++)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
                              "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s10", "s11");
    }
    if constexpr (OP == 13) { // VALU interleaved with SALU: do they issue in parallel?
        for (int i = 0; i < iters; i++)
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n s_and_b64 s[10:11], s[10:11], exec\n v_mul_f32 %1, %1, %8\n s_and_b64 s[10:11], s[10:11], exec\n"
                              "v_mul_f32 %2, %2, %8\n s_and_b64 s[10:11], s[10:11], exec\n v_mul_f32 %3, %3, %8\n s_and_b64 s[10:11], s[10:11], exec\n"
                              "v_mul_f32 %4, %4, %8\n s_and_b64 s[10:11], s[10:11], exec\n v_mul_f32 %5, %5, %8\n s_and_b64 s[10:11], s[10:11], exec\n"
                              "v_mul_f32 %6, %6, %8\n s_and_b64 s[10:11], s[10:11], exec\n v_mul_f32 %7, %7, %8\n s_and_b64 s[10:11], s[10:11], exec\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s10", "s11", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP> void run(const char *name, float *out, int n_cu) {
    const int iters = 4000;
    for (int block : {256, 1024}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(rate<OP>, dim3(n_cu), dim3(block), 0, 0, out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double instr = (double)iters * 64 * (block / 256.0);
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, block / 256,
               ms * 1e6 / instr * 2.4);
    }
}

int main() {
    int n_cu = 0;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
    run<11>("v_fma_f32", out, n_cu);
    run<0>("v_mul_f32", out, n_cu);
    run<1>("v_min_f32", out, n_cu);
    run<5>("v_max3_f32", out, n_cu);
    run<2>("v_add_u32", out, n_cu);
    run<3>("v_and_b32", out, n_cu);
    run<10>("v_mov_b32", out, n_cu);
    run<4>("v_cndmask_b32 (vcc)", out, n_cu);
    run<12>("v_cndmask_b32 (sgpr pair)", out, n_cu);
    run<6>("v_cmp_lt_f32 -> vcc", out, n_cu);
    run<7>("v_rcp_f32", out, n_cu);
    run<8>("v_sqrt_f32", out, n_cu);
    run<9>("v_sin_f32", out, n_cu);
    run<13>("v_mul_f32 + s_and_b64 pairs (per pair)", out, n_cu);
    return 0;
}
