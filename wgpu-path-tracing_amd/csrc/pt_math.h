// pt_math.h — device-side arithmetic contract.
//
// Every helper names the exact IEEE-754 binary32 operation order the kernels use.
// The translation unit is compiled with -ffp-contract=off -fno-fast-math, so the only
// fused multiply-adds are the explicit __builtin_fmaf below; '/' and sqrtf are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt);
// min/max are v_min_f32 / v_max_f32 (NaN -> the other operand, max(+0,-0) = +0,
// min(+0,-0) = -0). The WGSL operators these stand for are in src/shader/pt.wgsl
// of the reference; the choices inside WGSL's accuracy envelope (reciprocal
// multiply for vector/scalar division and the slab test, fused dot/cross/linear
// combinations, pow5, polynomial sin/cos) are listed in DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_PI  3.14159265359f     // pt.wgsl:3
#define PT_EPS 1e-6f              // pt.wgsl:4
#define PT_DEV __device__ __forceinline__

// Streams that are written once and read once, a kernel later (ray state, hit and shadow records): with PT_NT = 1 they are
// stored / loaded with the non-temporal hint, so that they do not displace scene data in the L2 (A/B switch; off by default).
#ifndef PT_NT
#define PT_NT 0
#endif
typedef float pt_f4n __attribute__((ext_vector_type(4)));
typedef float pt_f2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(float4 *p, float4 v) {
#if PT_NT
    pt_f4n n = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(n, reinterpret_cast<pt_f4n *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(float2 *p, float2 v) {
#if PT_NT
    pt_f2n n = {v.x, v.y}; __builtin_nontemporal_store(n, reinterpret_cast<pt_f2n *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
#if PT_NT
    pt_f4n n = __builtin_nontemporal_load(reinterpret_cast<const pt_f4n *>(p)); return make_float4(n.x, n.y, n.z, n.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ float2 ld_stream(const float2 *p) {
#if PT_NT
    pt_f2n n = __builtin_nontemporal_load(reinterpret_cast<const pt_f2n *>(p)); return make_float2(n.x, n.y);
#else
    return *p;
#endif
}

struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };

// ---- correctly rounded 1/x and sqrt(x) in fewer instructions ---------------------------------------------------------
// The compiler's IEEE expansions take 11 instructions for 1.0f / x (two v_div_scale, v_rcp, five fma/mul, v_div_fmas,
// v_div_fixup) and about as many for sqrtf: they also cover denormal, huge and special operands. For operands with
// 2^-100 <= |x| <= 2^100 the sequences below give the SAME BITS on this hardware for every one of the 2^32 float patterns in that
// range — checked exhaustively, not sampled (tools/ubench/exact_math.hip; tests/test_gpu_math.py runs the library's own
// functions over all 2^32 inputs through ptmi_debug_exact_math) — so the arithmetic contract is unchanged: 1/x and sqrt(x)
// stay the correctly rounded IEEE results the CPU oracle computes with '/' and sqrtf. Outside that range (and for NaN) the
// IEEE expansion runs; a wave takes that branch only when one of its lanes needs it.
//   1/x:     r = v_rcp_f32(x) (1 ulp); e = fma(-x, r, 1); r + e r          (3 instructions)
//   sqrt(x): r = v_rsq_f32(x); s = x r; h = r / 2; e = fma(-s, s, x); s + e h   (5 instructions)
#ifndef PT_IEEE_EXPANSIONS
#define PT_IEEE_EXPANSIONS 0        /* 1: always the compiler's expansions (A/B and the reference side of the exhaustive test) */
#endif
PT_DEV float rcp_short(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
PT_DEV float sqrt_short(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r, h = 0.5f * r;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
// the short form runs unconditionally; the lanes outside its range (almost never any: one scalar branch) redo it the IEEE way
PT_DEV float rcp1(float x) {
#if PT_IEEE_EXPANSIONS
    return 1.0f / x;
#else
    const float ax = __builtin_fabsf(x);
    float r = rcp_short(x);
    if (__builtin_expect(!((ax >= 0x1p-100f) & (ax <= 0x1p100f)), 0)) r = 1.0f / x;
    return r;
#endif
}
// 1/x where the caller discards the result for |x| < 1e-6 anyway (tri_test): only the upper end needs the IEEE branch
PT_DEV float rcp1_above_eps(float x) {
#if PT_IEEE_EXPANSIONS
    return 1.0f / x;
#else
    float r = rcp_short(x);
    if (__builtin_expect(!(__builtin_fabsf(x) <= 0x1p100f), 0)) r = 1.0f / x;
    return r;
#endif
}
PT_DEV float sqrt1(float x) {
#if PT_IEEE_EXPANSIONS
    return __builtin_sqrtf(x);
#else
    float r = sqrt_short(x);
    if (__builtin_expect(!((x >= 0x1p-100f) & (x <= 0x1p100f)), 0)) r = __builtin_sqrtf(x);
    return r;
#endif
}

PT_DEV v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV v3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }
PT_DEV float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float min1(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float max1(float a, float b) { return __builtin_fmaxf(a, b); }
PT_DEV v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 mul3(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV v3 scale3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV v3 neg3(v3 a) { return mk3(-a.x, -a.y, -a.z); }
// (ax*bx + ay*by) + az*bz, each step fused
PT_DEV float dot3(v3 a, v3 b) { return fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x)); }
PT_DEV v3 cross3(v3 a, v3 b) {
    return mk3(fma1(a.y, b.z, -(a.z * b.y)), fma1(a.z, b.x, -(a.x * b.z)), fma1(a.x, b.y, -(a.y * b.x)));
}
PT_DEV v3 madd3(v3 a, float s, v3 b) { return mk3(fma1(a.x, s, b.x), fma1(a.y, s, b.y), fma1(a.z, s, b.z)); }
PT_DEV v3 lincomb3(v3 a, float s1, v3 b, float s2, v3 c, float s3) {
    return mk3(fma1(c.x, s3, fma1(b.x, s2, a.x * s1)), fma1(c.y, s3, fma1(b.y, s2, a.y * s1)),
               fma1(c.z, s3, fma1(b.z, s2, a.z * s1)));
}
// WGSL vector / scalar: one IEEE reciprocal, three multiplies
PT_DEV v3 vdiv3(v3 a, float s) { float inv = rcp1(s); return mk3(a.x * inv, a.y * inv, a.z * inv); }
PT_DEV float length3(v3 a) { return sqrt1(dot3(a, a)); }
// a / sqrt(a.a): one range test covers both short forms (a.a within [2^-100, 2^100] puts its root within [2^-50, 2^50])
PT_DEV v3 normalize3(v3 a) {
#if PT_IEEE_EXPANSIONS
    return vdiv3(a, length3(a));
#else
    const float l2 = dot3(a, a);
    float inv = rcp_short(sqrt_short(l2));
    if (__builtin_expect(!((l2 >= 0x1p-100f) & (l2 <= 0x1p100f)), 0)) inv = 1.0f / __builtin_sqrtf(l2);
    return mk3(a.x * inv, a.y * inv, a.z * inv);
#endif
}
PT_DEV float mix1(float a, float b, float t) { return fma1(b, t, a * (1.0f - t)); }
PT_DEV v3 reflect3(v3 i, v3 n) {
    float k = 2.0f * dot3(n, i);
    return mk3(fma1(-k, n.x, i.x), fma1(-k, n.y, i.y), fma1(-k, n.z, i.z));
}
PT_DEV v3 refract3(v3 i, v3 n, float eta) {
    float dn = dot3(n, i);
    float k = 1.0f - (eta * eta) * (1.0f - dn * dn);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float c = eta * dn + sqrt1(k);
    return mk3(fma1(-c, n.x, eta * i.x), fma1(-c, n.y, eta * i.y), fma1(-c, n.z, eta * i.z));
}
PT_DEV float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// sin/cos on [0, ~2*pi]: quadrant reduction + fixed polynomials (same constants,
// same fused steps as the oracle's contract build).
PT_DEV void sincos1(float x, float &s, float &c) {
    const float TWO_OVER_PI = 0.636619772f;
    const float PIO2_HI = 1.57079637f;
    const float PIO2_LO = -4.37113883e-08f;
    float fk = __builtin_floorf(x * TWO_OVER_PI + 0.5f);
    float r = fma1(-fk, PIO2_HI, x);
    r = fma1(-fk, PIO2_LO, r);
    float r2 = r * r;
    float ps = fma1(-1.95152959e-4f, r2, 8.33216087e-3f);
    ps = fma1(ps, r2, -1.66666546e-1f);
    ps = fma1(ps, r2 * r, r);
    float pc = fma1(2.44331571e-5f, r2, -1.38873163e-3f);
    pc = fma1(pc, r2, 4.16666457e-2f);
    pc = fma1(pc, r2 * r2, fma1(-0.5f, r2, 1.0f));
    int q = (int)fk & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
PT_DEV float tan1(float x) { float s, c; sincos1(x, s, c); return s / c; }

// WGSL u32(f): truncating, saturating, NaN -> 0
PT_DEV uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// Moller-Trumbore, pt.wgsl:128-158; returns t (> 1e-6) or -1. Straight-line: the reference's four early returns
// (:134, :143, :151, :157) are folded into one predicate with the same NaN behaviour (a NaN never
// triggers an early return there, and fails the final t > EPSILON here as there). e1 = v1 - v0, e2 = v2 - v0.
// Used by the traversal kernels for every candidate and by `shade` to rebuild (u, v) of the closest hit from its
// triangle (the hit record carries only t and the triangle): same function, same operands, same bits.
// BOUNDED: the caller knows |a| <= 2^100 for this ray and every triangle of the scene (traverse.hip: DevScene::tri_safe_dsum),
// so the short reciprocal needs no range test — the triangle loop stays one basic block. Same bits either way.
template <bool BOUNDED>
PT_DEV float tri_test_t(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float &uo, float &vo) {
    v3 h = cross3(d, e2);
    float a = dot3(e1, h);
    float f = (BOUNDED && !PT_IEEE_EXPANSIONS) ? rcp_short(a) : rcp1_above_eps(a);       // |a| < PT_EPS is rejected below whatever f is
    v3 sv = sub3(o, v0);
    float u = f * dot3(sv, h);
    v3 q = cross3(sv, e1);
    float v = f * dot3(d, q);
    float t = f * dot3(e2, q);
    bool reject = (__builtin_fabsf(a) < PT_EPS) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
    bool ok = !reject & (t > PT_EPS);
    uo = u; vo = v;
    return ok ? t : -1.0f;
}
PT_DEV float tri_test(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float &uo, float &vo) { return tri_test_t<false>(v0, e1, e2, o, d, uo, vo); }

// ---- RNG: src/shader/random.wgsl:3-16 ---------------------------------------
PT_DEV uint32_t rng_seed(uint32_t x, uint32_t y, uint32_t frame) { return x + y * 1000u + frame * 100000u; }
PT_DEV uint32_t rng_word(uint32_t &st) {
    uint32_t s = st * 747796405u + 2891336453u;
    st = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}
// f32(word) / f32(4294967295.0): the divisor is 2^32 in f32, the quotient exact (can be 1.0)
PT_DEV float rng_f(uint32_t &st) { return (float)rng_word(st) * 2.3283064365386963e-10f; }
PT_DEV uint32_t rng_int(uint32_t &st, uint32_t lo, uint32_t hi) {
    uint32_t span = hi - lo + 1u;
    uint32_t k = f2u(rng_f(st) * (float)span);
    if (k > hi - lo) k = hi - lo;       // rand()==1.0 would index one past the end
    return lo + k;
}
