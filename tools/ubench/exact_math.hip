// tools/ubench/exact_math.hip — exhaustive check, over ALL 2^32 float bit patterns, of short instruction sequences for 1/x and
// sqrt(x) against the compiler's IEEE expansions (-fno-fast-math): which inputs, if any, give different bits?
//   hipcc -O3 --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -o tools/ubench/exact_math.bin tools/ubench/exact_math.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

__device__ __forceinline__ float rcp_nr1(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_nr2(float x) {
    float r = rcp_nr1(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_nr1(float x) {            // s = x * rsq(x), one correction with h = rsq / 2
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_nr2(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_hw1(float x) {            // the hardware's sqrt (1 ulp) + one correction with 1 / (2 s)
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

// counters: per candidate [all inputs, inputs with 2^-100 <= |x| <= 2^100], first differing input
__global__ void k_check(unsigned long long *cnt, uint32_t *first) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ax = __builtin_fabsf(x);
        const bool mid = ax >= 0x1p-100f && ax <= 0x1p100f;
        const float ref_r = 1.0f / x, ref_s = __builtin_sqrtf(x);
        const float c[5] = {rcp_nr1(x), rcp_nr2(x), sqrt_nr1(x), sqrt_nr2(x), sqrt_hw1(x)};
        for (int k = 0; k < 5; k++) {
            const float ref = k < 2 ? ref_r : ref_s;
            const bool same = __float_as_uint(c[k]) == __float_as_uint(ref) || (c[k] != c[k] && ref != ref);
            const bool dom = k < 2 ? mid : (mid && x > 0.0f);
            if (!same) {
                atomicAdd(&cnt[2 * k], 1ull);
                if (dom) { atomicAdd(&cnt[2 * k + 1], 1ull); atomicMin(&first[k], bits); }
            }
        }
    }
}

int main() {
    unsigned long long *cnt; uint32_t *first;
    hipMalloc(&cnt, 10 * sizeof *cnt); hipMalloc(&first, 5 * sizeof *first);
    hipMemset(cnt, 0, 10 * sizeof *cnt); hipMemset(first, 0xFF, 5 * sizeof *first);
    hipLaunchKernelGGL(k_check, dim3(256 * 16), dim3(256), 0, 0, cnt, first);
    unsigned long long h[10]; uint32_t f[5];
    if (hipMemcpy(h, cnt, sizeof h, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(f, first, sizeof f, hipMemcpyDeviceToHost) != hipSuccess) { puts("failed"); return 1; }
    const char *name[5] = {"rcp + 1 Newton step", "rcp + 2 Newton steps", "rsq: s = x r, 1 correction", "rsq: 2 corrections", "sqrt + 1 correction"};
    for (int k = 0; k < 5; k++)
        printf("%-28s differs from IEEE on %llu of 2^32 inputs, %llu of those with 2^-100 <= |x| <= 2^100%s (first 0x%08x)\n", name[k], h[2 * k], h[2 * k + 1],
               k < 2 ? "" : ", x > 0", f[k]);
    return 0;
}
