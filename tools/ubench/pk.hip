#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float b = 1.0001f, c = 0.5f; float2v bb = {b, b}, cc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb), "v"(cc));
        } else if (MODE == 2) {
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb), "v"(cc));
        } else {
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                         "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min3_f32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 123.456f) out[0] = r;
}
template <int MODE> void run(const char *name, int blocks, float *d) {
    int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 8;   // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9;       // at nominal 2.4 GHz
    double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-28s blocks=%5d waves/SIMD=%.0f  %.3f ms  cycles(2.4GHz)/wave-instr/SIMD = %.2f\n", name, blocks, waves_per_simd, ms,
           per_simd_cycles / (winstr / 1024.0));
}
int main() {
    float *d; hipMalloc(&d, 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("v_fma_f32", blocks, d); run<1>("v_pk_fma_f32", blocks, d); run<2>("v_pk_mul/add_f32", blocks, d); run<3>("mul/add/min/max/min3/cnd", blocks, d);
    }
    return 0;
}
