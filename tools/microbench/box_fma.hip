// Issue cost of the slab test's FMAs on gfx950, as scalar v_fma_f32 with abs / neg modifiers
// (what the node step uses) and as v_pk_fma_f32 over pairs of boxes with op_sel broadcasts of the
// per-ray constants.  Both variants do the arithmetic of four boxes per round (36 FMAs / 18
// packed FMAs) on values that stay in registers.  Build: hipcc --offload-arch=gfx950 -O3 box_fma.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(1024) rate(float *out, int iters) {
    const float t = threadIdx.x * 1e-3f;
    float inv0 = 1.0f + t, inv1 = 1.1f + t, inv2 = 1.2f + t, oi0 = 0.1f, oi1 = 0.2f, oi2 = 0.3f;
    v2f P1 = {inv0, inv1}, P2 = {inv2, oi0}, P3 = {oi1, oi2}, A1 = {inv0, inv1}, A2 = {inv2, 0.0f};
    float c[12], h[12];
    v2f pc[6], ph[6];
    for (int i = 0; i < 12; i++) c[i] = t + i, h[i] = 0.5f + 0.01f * i;
    for (int i = 0; i < 6; i++) pc[i] = v2f{c[2 * i], c[2 * i + 1]}, ph[i] = v2f{h[2 * i], h[2 * i + 1]};
    float acc = 0.0f;
    v2f pacc = {0.0f, 0.0f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                float mx, my, mz, nx, ny, nz, fx, fy, fz;
                asm volatile("v_fma_f32 %0, %9, %12, %15\n v_fma_f32 %1, %10, %13, %16\n v_fma_f32 %2, %11, %14, %17\n"
                             "v_fma_f32 %3, -%18, |%12|, %0\n v_fma_f32 %4, %18, |%12|, %0\n"
                             "v_fma_f32 %5, -%19, |%13|, %1\n v_fma_f32 %6, %19, |%13|, %1\n"
                             "v_fma_f32 %7, -%20, |%14|, %2\n v_fma_f32 %8, %20, |%14|, %2\n"
                             : "=&v"(mx), "=&v"(my), "=&v"(mz), "=&v"(nx), "=&v"(fx), "=&v"(ny), "=&v"(fy), "=&v"(nz), "=&v"(fz)
                             : "v"(c[3 * b]), "v"(c[3 * b + 1]), "v"(c[3 * b + 2]), "v"(inv0), "v"(inv1), "v"(inv2), "v"(oi0),
                               "v"(oi1), "v"(oi2), "v"(h[3 * b]), "v"(h[3 * b + 1]), "v"(h[3 * b + 2]));
                acc += nx + fx + ny + fy + nz + fz;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                v2f mx, my, mz, nx, ny, nz, fx, fy, fz;
                asm volatile("v_pk_fma_f32 %0, %9, %12, %13 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %1, %10, %12, %14 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n"
                             "v_pk_fma_f32 %2, %11, %13, %14 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %3, %17, %15, %0 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                             "v_pk_fma_f32 %4, %17, %15, %0 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %5, %18, %15, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                             "v_pk_fma_f32 %6, %18, %15, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                             "v_pk_fma_f32 %7, %19, %16, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                             "v_pk_fma_f32 %8, %19, %16, %2 op_sel_hi:[1,0,1]\n"
                             : "=&v"(mx), "=&v"(my), "=&v"(mz), "=&v"(nx), "=&v"(fx), "=&v"(ny), "=&v"(fy), "=&v"(nz), "=&v"(fz)
                             : "v"(pc[3 * p]), "v"(pc[3 * p + 1]), "v"(pc[3 * p + 2]), "v"(P1), "v"(P2), "v"(P3), "v"(A1), "v"(A2),
                               "v"(ph[3 * p]), "v"(ph[3 * p + 1]), "v"(ph[3 * p + 2]));
                pacc += nx + fx + ny + fy + nz + fz;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + pacc.x + pacc.y;
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    hipMalloc(&out, sizeof(float) * 1024 * 1024);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0)
                hipLaunchKernelGGL(rate<0>, dim3(n_cu), dim3(1024), 0, 0, out, iters);
            else
                hipLaunchKernelGGL(rate<1>, dim3(n_cu), dim3(1024), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // per round of four boxes: the FMAs plus the 20 (scalar) / 10 (packed) additions that keep the results alive
        printf("%s: %.3f ms for %d rounds of four boxes at 4 waves per SIMD -> %.1f ns per round per wave\n",
               mode ? "v_pk_fma_f32 x18" : "v_fma_f32 x36   ", ms, iters, ms * 1e6 / iters / 4);
    }
    return 0;
}
