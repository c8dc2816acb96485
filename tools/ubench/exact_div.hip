// tools/ubench/exact_div.hip — how often does a short quotient differ from the IEEE expansion?  a / b by
//   y = 1/b (correctly rounded: rcp + one correction, tools/ubench/exact_math.hip), q = a y, r = fma(-b, q, a), q + r y   [1 correction]
//   ... and once more: r' = fma(-b, q', a), q' + r' y                                                            [2 corrections]
// on pseudo-random operand pairs (all mantissas, exponents within +-40 so that nothing over- or underflows) and on structured
// ones (mantissas near all-ones / all-zeros). A sample, not a proof: two operands cannot be enumerated.
//   hipcc -O3 --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -o tools/ubench/exact_div.bin tools/ubench/exact_div.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp_short(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float make(uint32_t h, uint32_t mode) {
    uint32_t man = h & 0x7FFFFFu;
    if (mode == 1) man |= 0x7FFF00u;                  // leading ones
    if (mode == 2) man &= 0x0000FFu;                  // leading zeros
    if (mode == 3) man = 0x7FFFFFu - (h & 0xFFu);     // all ones minus a little
    const uint32_t e = 127u - 40u + (mix(h) % 81u);
    return __uint_as_float(((h >> 31) << 31) | (e << 23) | man);
}

__global__ void k_div(uint64_t n, uint32_t seed, unsigned long long *cnt) {
    unsigned long long bad1 = 0, bad2 = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h1 = mix((uint32_t)i ^ seed), h2 = mix((uint32_t)(i >> 32) * 0x9E3779B9u + h1 + 0x68E31DA4u);
        const uint32_t mode = (uint32_t)(i & 15u);
        const float a = make(h1, mode & 3u), b = make(h2, mode >> 2);
        const float ref = a / b;
        const float y = rcp_short(b);
        float q = a * y;
        q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
        const float q1 = q;
        q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
        bad1 += __float_as_uint(q1) != __float_as_uint(ref);
        bad2 += __float_as_uint(q) != __float_as_uint(ref);
    }
    if (bad1) atomicAdd(&cnt[0], bad1);
    if (bad2) atomicAdd(&cnt[1], bad2);
}

int main() {
    unsigned long long *cnt, h[2];
    if (hipMalloc(&cnt, sizeof h) != hipSuccess || hipMemset(cnt, 0, sizeof h) != hipSuccess) return 1;
    const uint64_t n = 1ull << 36;
    for (uint32_t round = 0; round < 4; round++) hipLaunchKernelGGL(k_div, dim3(256 * 16), dim3(256), 0, 0, n, 0x1234567u * (round + 1), cnt);
    if (hipMemcpy(h, cnt, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("pairs %llu: one correction differs on %llu, two corrections on %llu\n", (unsigned long long)(4 * n), h[0], h[1]);
    return 0;
}
