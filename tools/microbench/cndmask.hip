// v_cndmask_b32 / v_cmp issue patterns on gfx950 (cycles per instruction per SIMD, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S S S S S S S S
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m)

template <int OP> __global__ void __launch_bounds__(1024) rate(float *out, int iters) {
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 100.0f;
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == 0) { // e32, implicit vcc, vcc never written in the loop
#define T(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n"
            asm volatile(REP8(R8(T)) ARGS : "vcc");
#undef T
        }
        if constexpr (OP == 1) { // e64 naming vcc
#define T(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n"
            asm volatile(REP8(R8(T)) ARGS : "vcc");
#undef T
        }
        if constexpr (OP == 2) { // cmp -> vcc, cndmask <- vcc pairs (count both)
#define T(k) "v_cmp_lt_f32_e32 vcc, %" #k ", %8\n v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n"
            asm volatile(REP8(R8(T)) ARGS : "vcc");
#undef T
        }
        if constexpr (OP == 3) { // cmp -> sgpr pair, cndmask <- sgpr pair
#define T(k) "v_cmp_lt_f32_e64 s[10:11], %" #k ", %8\n v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
            asm volatile(REP8(R8(T)) ARGS : "s10", "s11");
#undef T
        }
        if constexpr (OP == 4) { // 4 cmps into 4 sgpr pairs, then 4 cndmasks
            asm volatile(REP8(
                "v_cmp_lt_f32_e64 s[10:11], %0, %8\n v_cmp_lt_f32_e64 s[12:13], %1, %8\n v_cmp_lt_f32_e64 s[14:15], %2, %8\n v_cmp_lt_f32_e64 s[16:17], %3, %8\n"
                "v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[12:13]\n v_cndmask_b32_e64 %2, %2, %8, s[14:15]\n v_cndmask_b32_e64 %3, %3, %8, s[16:17]\n"
                "v_cmp_lt_f32_e64 s[10:11], %4, %8\n v_cmp_lt_f32_e64 s[12:13], %5, %8\n v_cmp_lt_f32_e64 s[14:15], %6, %8\n v_cmp_lt_f32_e64 s[16:17], %7, %8\n"
                "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[12:13]\n v_cndmask_b32_e64 %6, %6, %8, s[14:15]\n v_cndmask_b32_e64 %7, %7, %8, s[16:17]\n")
                ARGS : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
        }
        if constexpr (OP == 5) { // s_and_saveexec / s_or exec pairs around one VALU instruction
#define T(k) "s_and_saveexec_b64 s[10:11], s[12:13]\n v_mul_f32 %" #k ", %" #k ", %8\n s_or_b64 exec, exec, s[10:11]\n"
            asm volatile("s_mov_b64 s[12:13], exec\n" REP8(R8(T)) ARGS : "s10", "s11", "s12", "s13", "scc");
#undef T
        }
        if constexpr (OP == 6) { // v_max_f32 / v_min_f32 alternating
#define T(k) "v_max_f32 %" #k ", %" #k ", %8\n"
            asm volatile(REP8(R8(T)) ARGS);
#undef T
        }
        if constexpr (OP == 7) { // v_med3 as clamp
#define T(k) "v_med3_f32 %" #k ", %" #k ", %8, %8\n"
            asm volatile(REP8(R8(T)) ARGS);
#undef T
        }
        if constexpr (OP == 8) { // v_add_f32
#define T(k) "v_add_f32 %" #k ", %" #k ", %8\n"
            asm volatile(REP8(R8(T)) ARGS);
#undef T
        }
        if constexpr (OP == 9) { // v_fma_f32 with three distinct VGPR sources
#define T(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
            asm volatile(REP8(R8(T)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(a0 * 0 + 1e-9f));
#undef T
        }
        if constexpr (OP == 10) { // v_fmac_f32 (VOP2)
#define T(k) "v_fmac_f32 %" #k ", %8, %8\n"
            asm volatile(REP8(R8(T)) ARGS);
#undef T
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char *name, float *out, int n_cu, int per) {
    const int iters = 3000;
    float ms = 0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate<OP>, dim3(n_cu), dim3(1024), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double instr = (double)iters * 64 * per * 4;
    printf("%-62s %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e6 / instr * 2.4);
}
int main() {
    int n_cu = 0;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
    run<8>("v_add_f32", out, n_cu, 1);
    run<9>("v_fma_f32 (3 distinct VGPRs)", out, n_cu, 1);
    run<10>("v_fmac_f32", out, n_cu, 1);
    run<6>("v_max_f32", out, n_cu, 1);
    run<7>("v_med3_f32", out, n_cu, 1);
    run<0>("v_cndmask_b32_e32 (vcc, never written)", out, n_cu, 1);
    run<1>("v_cndmask_b32_e64 (vcc, never written)", out, n_cu, 1);
    run<2>("v_cmp_e32 -> vcc + v_cndmask_e32 <- vcc (per instruction)", out, n_cu, 2);
    run<3>("v_cmp_e64 -> s[10:11] + v_cndmask_e64 <- s[10:11] (per instr.)", out, n_cu, 2);
    run<4>("4 v_cmp_e64 -> 4 sgpr pairs, then 4 v_cndmask_e64 (per instr.)", out, n_cu, 2);
    run<5>("s_and_saveexec + v_mul + s_or exec (per triple)", out, n_cu, 1);
    return 0;
}
