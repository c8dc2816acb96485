/*
 * pt_oracle.c — CPU oracle for the path-tracing hot path (TEST INFRASTRUCTURE,
 * see pt_oracle.h for the rules on who may use it and for the "parity unpinned"
 * statement).
 *
 * Scalar f32 restatement of the reference's WGSL compute shader, function by
 * function; every function cites the reference lines it follows
 * (pt.wgsl = /root/reference/src/shader/pt.wgsl, random.wgsl likewise).
 *
 * Two builds of this one file:
 *   PT_STRICT=0 (libpt_oracle.so)        the ARITHMETIC CONTRACT of DESIGN.md §3 —
 *       the exact operation order the HIP kernels implement, so that GPU and
 *       oracle agree bit for bit. Inside WGSL's accuracy envelope it picks:
 *       dot/cross/linear-combination helpers fused (fmaf), vector/scalar
 *       division as multiplication by the IEEE reciprocal, the slab test as
 *       (bound - o) * (1/d), pow(x,5) as x^2*x^2*x, sin/cos by a fixed
 *       polynomial. Everything else is one IEEE-754 binary32 operation per
 *       WGSL operator, never contracted (-ffp-contract=off).
 *   PT_STRICT=1 (libpt_oracle_strict.so) the literal transcription: every WGSL
 *       '/' is an IEEE division, no fused multiply-add anywhere, libm
 *       sinf/cosf/tanf/powf. tests/test_oracle.py::test_contract_vs_literal_build
 *       shows the two builds agree to Monte-Carlo-branch-flip level, and
 *       tests/test_gpu_strict.py compares the HIP path with THIS build directly.
 *
 * Compile: see oracle/Makefile (gcc -O2 -std=c11 -fopenmp -mfma -ffp-contract=off).
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef PT_STRICT
#define PT_STRICT 0
#endif

#define PT_PI   3.14159265359f          /* pt.wgsl:3 (rounds to 0x40490FDB) */
#define PT_EPS  1e-6f                   /* pt.wgsl:4 */
#define PT_STACK_MAX 1024               /* reference: 64, unguarded (pt.wgsl:249) */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

/* ------------------------------------------------------------------------- */
/* arithmetic contract helpers                                                */
/* ------------------------------------------------------------------------- */
static inline float fma_(float a, float b, float c) {
#if PT_STRICT
    return a * b + c;                   /* two roundings (-ffp-contract=off) */
#else
    return __builtin_fmaf(a, b, c);
#endif
}
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* dot(a,b) = (ax*bx + ay*by) + az*bz */
static inline float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(fma_(a.y, b.z, -(a.z * b.y)),
              fma_(a.z, b.x, -(a.x * b.z)),
              fma_(a.x, b.y, -(a.y * b.x)));
}
/* a*s + b */
static inline v3 madd3(v3 a, float s, v3 b) {
    return V3(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z));
}
/* a*s1 + b*s2 + c*s3, left to right */
static inline v3 lincomb3(v3 a, float s1, v3 b, float s2, v3 c, float s3) {
    return V3(fma_(c.x, s3, fma_(b.x, s2, a.x * s1)),
              fma_(c.y, s3, fma_(b.y, s2, a.y * s1)),
              fma_(c.z, s3, fma_(b.z, s2, a.z * s1)));
}
/* WGSL vector / scalar */
static inline v3 vdiv3(v3 a, float s) {
#if PT_STRICT
    return V3(a.x / s, a.y / s, a.z / s);
#else
    float inv = 1.0f / s;
    return V3(a.x * inv, a.y * inv, a.z * inv);
#endif
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { return vdiv3(a, length3(a)); }
/* WGSL mix(a,b,t) = a*(1-t) + b*t */
static inline float mix1(float a, float b, float t) { return fma_(b, t, a * (1.0f - t)); }
/* WGSL reflect(i,n) = i - 2*dot(n,i)*n */
static inline v3 reflect3(v3 i, v3 n) {
    float k = 2.0f * dot3(n, i);
    return V3(fma_(-k, n.x, i.x), fma_(-k, n.y, i.y), fma_(-k, n.z, i.z));
}
/* WGSL refract(i,n,eta) */
static inline v3 refract3(v3 i, v3 n, float eta) {
    float dn = dot3(n, i);
    float k = 1.0f - (eta * eta) * (1.0f - dn * dn);
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    float c = eta * dn + sqrtf(k);
    return V3(fma_(-c, n.x, eta * i.x), fma_(-c, n.y, eta * i.y), fma_(-c, n.z, eta * i.z));
}
/* pow(x, 5.0) (pt.wgsl:344, :619) */
static inline float pow5(float x) {
#if PT_STRICT
    return powf(x, 5.0f);
#else
    float x2 = x * x;
    return (x2 * x2) * x;
#endif
}
/* WGSL min/max: the NaN and signed-zero cases are implementation-defined there;
 * the contract takes the gfx950 v_max_f32 / v_min_f32 results: a NaN operand
 * yields the other operand, max(+0,-0) = +0, min(+0,-0) = -0. */
static inline float max1(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return signbit(a) ? b : a;
    return a >= b ? a : b;
}
static inline float min1(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return signbit(a) ? a : b;
    return a <= b ? a : b;
}

/* sin/cos for x in [0, ~2*pi]: quadrant reduction (two-term Cody-Waite) and
 * degree-7/8 polynomials on [-pi/4, pi/4]; always hardware-fused in the
 * contract build. */
static inline void sincos1(float x, float *s, float *c) {
#if PT_STRICT
    *s = sinf(x); *c = cosf(x);
#else
    const float TWO_OVER_PI = 0.636619772f;
    const float PIO2_HI = 1.57079637f;            /* RN(pi/2)            */
    const float PIO2_LO = -4.37113883e-08f;       /* pi/2 - PIO2_HI      */
    float fk = floorf(x * TWO_OVER_PI + 0.5f);
    float r = __builtin_fmaf(-fk, PIO2_HI, x);
    r = __builtin_fmaf(-fk, PIO2_LO, r);
    float r2 = r * r;
    float ps = __builtin_fmaf(-1.95152959e-4f, r2, 8.33216087e-3f);
    ps = __builtin_fmaf(ps, r2, -1.66666546e-1f);
    ps = __builtin_fmaf(ps, r2 * r, r);
    float pc = __builtin_fmaf(2.44331571e-5f, r2, -1.38873163e-3f);
    pc = __builtin_fmaf(pc, r2, 4.16666457e-2f);
    pc = __builtin_fmaf(pc, r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    int q = (int)fk & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
#endif
}
static inline float tan1(float x) {
#if PT_STRICT
    return tanf(x);
#else
    float s, c; sincos1(x, &s, &c); return s / c;
#endif
}
/* f32 -> u32, truncating and saturating (WGSL u32(f), vec2u(v)); NaN -> 0 */
static inline uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static inline float half2float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                   /* subnormal */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

/* ------------------------------------------------------------------------- */
/* RNG — random.wgsl:1-16                                                     */
/* ------------------------------------------------------------------------- */
uint32_t pto_seed(uint32_t x, uint32_t y, uint32_t frame) {            /* :3-5 */
    return x + y * 1000u + frame * 100000u;
}
static inline uint32_t rng_word(uint32_t *st) {                         /* :8-10 */
    uint32_t s = *st * 747796405u + 2891336453u;
    *st = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}
/* :11 — f32(result) / 4294967295.0; the divisor rounds to 2^32 in f32, so the
 * quotient is exact and can be 1.0 */
static inline float rng_f(uint32_t *st) { return (float)rng_word(st) / 4294967296.0f; }
static inline uint32_t rng_int(uint32_t *st, uint32_t lo, uint32_t hi) { /* :14-16 */
    uint32_t span = hi - lo + 1u;
    uint32_t k = f2u(rng_f(st) * (float)span);
    if (k > hi - lo) k = hi - lo;        /* rand()==1.0 would index one past the end */
    return lo + k;
}
void pto_rand(uint32_t *state_io, uint32_t n, uint32_t *states, uint32_t *words, float *vals) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t w = rng_word(state_io);
        if (states) states[i] = *state_io;
        if (words) words[i] = w;
        if (vals) vals[i] = (float)w / 4294967296.0f;
    }
}
uint32_t pto_rand_int(uint32_t *state_io, uint32_t lo, uint32_t hi) { return rng_int(state_io, lo, hi); }
int pto_is_strict(void) { return PT_STRICT; }
void pto_sincos(float x, float *s, float *c) { sincos1(x, s, c); }

/* ------------------------------------------------------------------------- */
/* types — pt.wgsl:80-101                                                     */
/* ------------------------------------------------------------------------- */
typedef struct { v3 o, d; } ray_t;
typedef struct { float t, u, v; uint32_t tri; } hit_t;          /* what leaves traversal */
typedef struct {                                                 /* pt.wgsl:86-101 */
    v3 position; float t; v3 normal; uint32_t material_index;
    v3 albedo; float alpha, roughness, metallic, transmission, ior;
    v3 emission; float emissive_strength; float uvx, uvy; int is_front;
} hitinfo_t;
/* ray tap (pto_render_tap): every traversal of a render, with its result, for tools that replay real rays elsewhere */
typedef struct { float *rec; uint64_t cap; uint64_t *n; uint64_t last; } ray_tap_t;
typedef struct {
    uint64_t segments, shadow_rays, nodes_visited, tris_tested, closest_hits;
    uint32_t max_stack;
    ray_tap_t *tap;
} counters_t;

/* ------------------------------------------------------------------------- */
/* textures — pt.wgsl:112-120                                                 */
/* ------------------------------------------------------------------------- */
static v4 atlas_load(const pto_scene *s, uint32_t x, uint32_t y) {
    v4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (!s->atlas || s->atlas_fmt == PTO_ATLAS_NONE || x >= s->atlas_w || y >= s->atlas_h)
        return z;                                    /* out of bounds reads zero */
    size_t idx = ((size_t)y * s->atlas_w + x) * 4;
    if (s->atlas_fmt == PTO_ATLAS_RGBA16F) {
        const uint16_t *p = (const uint16_t *)s->atlas + idx;
        v4 r = { half2float(p[0]), half2float(p[1]), half2float(p[2]), half2float(p[3]) };
        return r;
    }
    const float *p = (const float *)s->atlas + idx;
    v4 r = { p[0], p[1], p[2], p[3] };
    return r;
}
static v4 texture_color(const pto_scene *s, const ptmi_atlas_rect *tx, float uvx, float uvy, v4 fallback) {
    if (tx->w == 0u || tx->h == 0u) return fallback;             /* :119 select() */
    float fx = uvx - truncf(uvx);                                 /* uv % 1.0, exact */
    float fy = uvy - truncf(uvy);
    float ax = (float)tx->x + fx * (float)tx->w;                  /* :115 */
    float ay = (float)tx->y + fy * (float)tx->h;                  /* :116 */
    return atlas_load(s, f2u(ax), f2u(ay));
}

/* ------------------------------------------------------------------------- */
/* intersection — pt.wgsl:123-158 (Moller-Trumbore part), :234-296             */
/* ------------------------------------------------------------------------- */
static inline float tri_test(const ptmi_triangle *T, v3 o, v3 d, float *uo, float *vo) {
    v3 v0 = ld3(T->v0);
    v3 e1 = sub3(ld3(T->v1), v0);                    /* :128 */
    v3 e2 = sub3(ld3(T->v2), v0);                    /* :129 */
    v3 h = cross3(d, e2);                            /* :130 */
    float a = dot3(e1, h);                           /* :131 */
    if (fabsf(a) < PT_EPS) return -1.0f;             /* :134 */
    float f = 1.0f / a;                              /* :138 */
    v3 sv = sub3(o, v0);                             /* :139 */
    float u = f * dot3(sv, h);                       /* :140 */
    if (u < 0.0f || u > 1.0f) return -1.0f;          /* :143 */
    v3 q = cross3(sv, e1);                           /* :147 */
    float v = f * dot3(d, q);                        /* :148 */
    if (v < 0.0f || u + v > 1.0f) return -1.0f;      /* :151 */
    float t = f * dot3(e2, q);                       /* :156 */
    if (t > PT_EPS) { *uo = u; *vo = v; return t; }  /* :157 */
    return -1.0f;
}

static inline int box_test(const ptmi_bvh_node *n, v3 o, v3 d, v3 inv_d) {     /* :234-245 */
#if PT_STRICT
    (void)inv_d;
    float t1x = (n->aabb_min[0] - o.x) / d.x, t2x = (n->aabb_max[0] - o.x) / d.x;
    float t1y = (n->aabb_min[1] - o.y) / d.y, t2y = (n->aabb_max[1] - o.y) / d.y;
    float t1z = (n->aabb_min[2] - o.z) / d.z, t2z = (n->aabb_max[2] - o.z) / d.z;
#else
    (void)d;
    float t1x = (n->aabb_min[0] - o.x) * inv_d.x, t2x = (n->aabb_max[0] - o.x) * inv_d.x;
    float t1y = (n->aabb_min[1] - o.y) * inv_d.y, t2y = (n->aabb_max[1] - o.y) * inv_d.y;
    float t1z = (n->aabb_min[2] - o.z) * inv_d.z, t2z = (n->aabb_max[2] - o.z) * inv_d.z;
#endif
    float tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= tmin && tmax >= 0.0f;
}

/* traverseBVH, pt.wgsl:248-291: DFS with an explicit stack, right pushed first
 * so left pops first; no distance cull; a hit replaces the closest only when
 * strictly nearer (first found wins ties). Only (t, tri, u, v) is kept here;
 * the shading state of :159-226 is a pure function of it (make_hitinfo). */
static hit_t traverse(const pto_scene *s, ray_t r, counters_t *c) {
    uint32_t stack[PT_STACK_MAX];
    uint32_t sp = 0;
    hit_t best = { -1.0f, 0.0f, 0.0f, 0xFFFFFFFFu };
    int has_hit = 0;
    v3 inv_d = V3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    if (s->n_nodes == 0) return best;
    stack[sp++] = 0u;
    while (sp > 0u) {
        uint32_t idx = stack[--sp];
        if (idx >= s->n_nodes) continue;             /* out of range reads nothing */
        const ptmi_bvh_node *n = &s->nodes[idx];
        c->nodes_visited++;
        if (!box_test(n, r.o, r.d, inv_d)) continue;                 /* :266 */
        if (n->triangle_count > 0u) {                                /* :271 */
            for (uint32_t i = 0; i < n->triangle_count; i++) {
                uint32_t ti = n->triangle_offset + i;
                if (ti >= s->n_tris) continue;
                float u = 0.0f, v = 0.0f;
                c->tris_tested++;
                float t = tri_test(&s->tris[ti], r.o, r.d, &u, &v);
                if (t > 0.0f && (t < best.t || !has_hit)) {          /* :275 */
                    best.t = t; best.u = u; best.v = v; best.tri = ti;
                    has_hit = 1;
                }
            }
        } else if (sp + 2 <= PT_STACK_MAX) {
            stack[sp++] = n->right;                                  /* :283 */
            stack[sp++] = n->left;                                   /* :285 */
            if (sp > c->max_stack) c->max_stack = sp;
        }
    }
    if (c->tap) {                                    /* 9 words per ray: o, d, dist (0: a closest-hit ray), t, tri */
        ray_tap_t *tp = c->tap;
        uint64_t i = __atomic_fetch_add(tp->n, 1, __ATOMIC_RELAXED);
        tp->last = i;
        if (i < tp->cap) {
            float *q = tp->rec + 9 * i;
            q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = r.d.x; q[4] = r.d.y; q[5] = r.d.z; q[6] = 0.0f;
            q[7] = best.t; memcpy(&q[8], &best.tri, 4);
        }
    }
    return best;
}
/* the ray just traced was a shadow ray: dist < 0 directional (pt.wgsl:392), else the distance of :421 / :463 */
static void tap_shadow(counters_t *c, float dist) {
    if (c->tap && c->tap->last < c->tap->cap) c->tap->rec[9 * c->tap->last + 6] = dist;
}

/* rayTriangleIntersect, pt.wgsl:159-226, for the winning triangle only */
static hitinfo_t make_hitinfo(const pto_scene *s, ray_t r, hit_t h) {
    hitinfo_t hi;
    memset(&hi, 0, sizeof hi);
    hi.t = h.t;
    if (h.t < 0.0f) return hi;
    const ptmi_triangle *T = &s->tris[h.tri];
    float u = h.u, v = h.v;
    v3 v0 = ld3(T->v0);
    v3 e1 = sub3(ld3(T->v1), v0), e2 = sub3(ld3(T->v2), v0);
    hi.position = madd3(r.d, h.t, r.o);                                      /* :159 */
    float w = 1.0f - u - v;                                                  /* :162 */
    v3 geo_n = normalize3(cross3(e1, e2));                                   /* :165 */
    v3 n_i = normalize3(lincomb3(ld3(T->n0), w, ld3(T->n1), u, ld3(T->n2), v)); /* :168-172 */
    hi.uvx = fma_(T->uv2[0], v, fma_(T->uv1[0], u, T->uv0[0] * w));          /* :192 */
    hi.uvy = fma_(T->uv2[1], v, fma_(T->uv1[1], u, T->uv0[1] * w));
    hi.material_index = T->material_index;                                   /* :193 */
    hi.is_front = dot3(geo_n, r.d) < 0.0f;                                   /* :196 */
    ptmi_material zero_m; memset(&zero_m, 0, sizeof zero_m);
    const ptmi_material *m = hi.material_index < s->n_mats ? &s->mats[hi.material_index] : &zero_m; /* :200 */
    v4 one = { 1.0f, 1.0f, 1.0f, 1.0f };
    v4 alb = texture_color(s, &m->albedo_map, hi.uvx, hi.uvy, one);          /* :203 */
    hi.albedo = V3(alb.x * m->base_color[0], alb.y * m->base_color[1], alb.z * m->base_color[2]);
    hi.alpha = alb.w;
    v4 pbr = texture_color(s, &m->pbr_map, hi.uvx, hi.uvy, one);             /* :206 */
    hi.metallic = pbr.z * m->metallic;                                       /* :207 */
    hi.roughness = max1(pbr.y * m->roughness, 0.04f);                        /* :208 */
    hi.transmission = m->transmission;
    hi.ior = m->ior;
    v4 em = texture_color(s, &m->emissive_map, hi.uvx, hi.uvy, one);         /* :211 */
    hi.emission = V3(em.x * m->emission[0], em.y * m->emission[1], em.z * m->emission[2]);
    hi.emissive_strength = m->emissive_strength;
    v4 flat = { 0.5f, 0.5f, 1.0f, 1.0f };
    v4 nm = texture_color(s, &m->normal_map, hi.uvx, hi.uvy, flat);          /* :216 */
    if (nm.x != 0.5f || nm.y != 0.5f || nm.z != 1.0f) {                      /* :217 */
        /* tangent frame, :176-189 (the bitangent of :183 is never used) */
        float du1x = T->uv1[0] - T->uv0[0], du1y = T->uv1[1] - T->uv0[1];
        float du2x = T->uv2[0] - T->uv0[0], du2y = T->uv2[1] - T->uv0[1];
        float rr = 1.0f / fma_(du1x, du2y, -(du1y * du2x));                  /* :181 */
        v3 tg = V3(fma_(e1.x, du2y, -(e2.x * du1y)) * rr,
                   fma_(e1.y, du2y, -(e2.y * du1y)) * rr,
                   fma_(e1.z, du2y, -(e2.z * du1y)) * rr);
        tg = normalize3(tg);                                                 /* :182 */
        v3 N = n_i;
        v3 Tn = normalize3(madd3(N, -dot3(N, tg), tg));                      /* :187 */
        v3 Bn = normalize3(cross3(N, Tn));                                   /* :188 */
        float tx = nm.x * 2.0f - 1.0f, ty = nm.y * 2.0f - 1.0f, tz = nm.z * 2.0f - 1.0f; /* :219 */
        hi.normal = normalize3(lincomb3(Tn, tx, Bn, ty, N, tz));             /* :222 */
    } else {
        hi.normal = n_i;                                                     /* :225 */
    }
    return hi;
}

static hitinfo_t scene_intersect(const pto_scene *s, ray_t r, counters_t *c, hit_t *raw) { /* :294-296 */
    hit_t h = traverse(s, r, c);
    if (raw) *raw = h;
    return make_hitinfo(s, r, h);
}

/* ------------------------------------------------------------------------- */
/* sampling and BSDF — pt.wgsl:299-364, :492-634                               */
/* ------------------------------------------------------------------------- */
static v3 random_cosine_direction(uint32_t *rng) {                  /* :299-307 */
    float r1 = rng_f(rng), r2 = rng_f(rng);
    float z = sqrtf(1.0f - r2);
    float phi = (2.0f * PT_PI) * r1;
    float sp, cp; sincos1(phi, &sp, &cp);
    float sr = sqrtf(r2);
    return V3(cp * sr, sp * sr, z);
}
static float distribution_ggx(v3 N, v3 H, float roughness) {        /* :316-325 */
    float a = roughness * roughness;
    float a2 = a * a;
    float ndh = max1(dot3(N, H), 0.0f);
    float ndh2 = ndh * ndh;
    float denom = ndh2 * (a2 - 1.0f) + 1.0f;
    return max1(a2 / (PT_PI * denom * denom), 0.0f);
}
static float geometry_schlick_ggx(float ndv, float roughness) {     /* :328-332 */
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return ndv / (ndv * (1.0f - k) + k);
}
static float geometry_smith(v3 N, v3 Vv, v3 L, float roughness) {   /* :334-340 */
    float ndv = max1(dot3(N, Vv), 0.0f);
    float ndl = max1(dot3(N, L), 0.0f);
    float g2 = geometry_schlick_ggx(ndv, roughness);
    float g1 = geometry_schlick_ggx(ndl, roughness);
    return g1 * g2;
}
static v3 fresnel_schlick(float cos_theta, v3 F0) {                 /* :343-345 */
    float p = pow5(1.0f - cos_theta);
    return V3(fma_(1.0f - F0.x, p, F0.x), fma_(1.0f - F0.y, p, F0.y), fma_(1.0f - F0.z, p, F0.z));
}
static float reflectance(float cos_theta, float eta) {              /* :616-620 */
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cos_theta);
}
/* constructTBN, :624-634: columns T, B, N */
static void construct_tbn(v3 N, v3 *To, v3 *Bo) {
    v3 T = V3(1.0f, 0.0f, 0.0f);
    if (fabsf(N.x) > 0.9f) T = V3(0.0f, 1.0f, 0.0f);
    v3 B = normalize3(cross3(N, T));
    T = normalize3(cross3(B, N));
    *To = T; *Bo = B;
}
static v3 sample_ggx_normal(uint32_t *rng, v3 normal, float roughness) {  /* :348-364 */
    float r1 = rng_f(rng), r2 = rng_f(rng);
    float a = roughness * roughness;
    float phi = (2.0f * PT_PI) * r1;
    float cos_t = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    float sp, cp; sincos1(phi, &sp, &cp);
    v3 T, B; construct_tbn(normal, &T, &B);
    return normalize3(lincomb3(T, sin_t * cp, B, sin_t * sp, normal, cos_t));
}
static float power_heuristic(float nf, float fpdf, float ng, float gpdf) {  /* :492-496 */
    float f = nf * fpdf, g = ng * gpdf;
    return (f * f) / (f * f + g * g);
}
/* sampleBSDF, :498-546 (direction is returned un-normalised, as there) */
static v3 sample_bsdf(uint32_t *rng, const hitinfo_t *h, ray_t cur, int front) {
    v3 Vv = neg3(normalize3(cur.d));                                /* :500 */
    float diffuse_p = (1.0f - h->metallic) * (1.0f - h->transmission);
    float specular_p = h->metallic;
    float r = rng_f(rng);                                           /* :508 */
    if (r < diffuse_p) {
        v3 l = random_cosine_direction(rng);
        v3 T, B; construct_tbn(h->normal, &T, &B);
        return lincomb3(T, l.x, B, l.y, h->normal, l.z);            /* :514 */
    } else if (r < diffuse_p + specular_p) {
        float rough = max1(h->roughness, 0.04f);
        v3 N = sample_ggx_normal(rng, h->normal, rough);
        return reflect3(neg3(Vv), N);                               /* :520 */
    } else {
        float eta = front ? 1.0f / h->ior : h->ior;                 /* :524 */
        float rough = max1(h->roughness, 0.04f);
        v3 N = sample_ggx_normal(rng, h->normal, rough);
        if (!front) N = neg3(N);                                    /* :528 */
        float cos_t = dot3(N, Vv);
        float sin_t = sqrtf(1.0f - cos_t * cos_t);
        int cannot_refract = eta * sin_t > 1.0f;
        float F = reflectance(fabsf(cos_t), eta);
        if (cannot_refract || (rng_f(rng) < F))                     /* :538 short-circuit */
            return reflect3(neg3(Vv), N);
        return refract3(neg3(Vv), N, eta);                          /* :543 */
    }
}
/* evalBSDF, :548-614: returns (f*cos, pdf) */
static v4 eval_bsdf(const hitinfo_t *h, v3 normal, v3 Vv, v3 L, int front) {
    v3 H = normalize3(add3(Vv, L));
    float ndl = max1(dot3(normal, L), 0.0f);
    float ndv = max1(dot3(normal, Vv), 0.0f);
    float ndh = max1(dot3(normal, H), 0.0f);
    float vdh = max1(dot3(Vv, H), 0.0f);
    v3 F0 = V3(mix1(0.04f, h->albedo.x, h->metallic), mix1(0.04f, h->albedo.y, h->metallic),
               mix1(0.04f, h->albedo.z, h->metallic));                       /* :559 */
    v3 F = fresnel_schlick(vdh, F0);
    float G = geometry_smith(normal, Vv, L, h->roughness);
    float D = distribution_ggx(normal, H, h->roughness);
    float one_m_tr = 1.0f - h->transmission;
    v3 kD = V3((1.0f - F.x) * one_m_tr, (1.0f - F.y) * one_m_tr, (1.0f - F.z) * one_m_tr); /* :571 */
    v3 diffuse = vdiv3(mul3(kD, h->albedo), PT_PI);                          /* :572 */
    v3 specular = vdiv3(scale3(scale3(F, G), D), max1(4.0f * ndv * ndl, PT_EPS)); /* :575 */
    v3 bsdf = V3(0.0f, 0.0f, 0.0f);
    float pdf = 0.0f;
    if (h->transmission > 0.0f) {                                            /* :581-594 */
        float eta = front ? 1.0f / h->ior : h->ior;
        float cos_t = dot3(normal, Vv);
        float Ft = reflectance(fabsf(cos_t), eta);
        bsdf = scale3(h->albedo, 1.0f - Ft);
        pdf = (1.0f - h->metallic) * h->transmission;
    } else {
        bsdf = scale3(add3(diffuse, specular), ndl);                         /* :597 */
        float diffuse_p = (1.0f - h->metallic) * (1.0f - h->transmission);
        float specular_p = h->metallic;
        float diffuse_pdf = ndl / PT_PI;
        float specular_pdf = D * ndh / (4.0f * vdh);
        pdf = diffuse_p * diffuse_pdf + specular_p * specular_pdf;           /* :610 */
    }
    v4 r = { bsdf.x, bsdf.y, bsdf.z, max1(pdf, PT_EPS) };                    /* :613 */
    return r;
}

/* ------------------------------------------------------------------------- */
/* light sampling — pt.wgsl:374-489                                            */
/* ------------------------------------------------------------------------- */
typedef struct { v3 intensity; uint32_t light_type; v3 wi; float pdf; } light_sample_t;

static light_sample_t sample_light(const pto_scene *s, uint32_t *rng, v3 hit_pos, counters_t *c) {
    light_sample_t ls;
    memset(&ls, 0, sizeof ls);
    uint32_t nl = s->n_lights;                       /* caller guarantees nl > 0 */
    const ptmi_light *lt = &s->lights[rng_int(rng, 0u, nl - 1u)];    /* :375 */
    ls.light_type = lt->light_type;
    float inv_n = 1.0f / (float)nl;
    if (lt->light_type == PTMI_LIGHT_DIRECTIONAL) {                   /* :385-406 */
        v3 wi = normalize3(neg3(ld3(lt->position)));
        ray_t sr = { madd3(wi, PT_EPS, hit_pos), wi };
        c->shadow_rays++;
        hit_t sh = traverse(s, sr, c);
        tap_shadow(c, -1.0f);
        if (sh.t > 0.0f) { ls.wi = wi; ls.pdf = 0.0f; return ls; }
        ls.intensity = scale3(ld3(lt->color), lt->intensity);
        ls.wi = wi;
        ls.pdf = inv_n * 1000.0f;
    } else if (lt->light_type == PTMI_LIGHT_POINT) {                  /* :407-438 */
        v3 to_l = sub3(ld3(lt->position), hit_pos);
        float dist = length3(to_l);
        if (dist > 100.0f) return ls;
        v3 wi = vdiv3(to_l, dist);
        ray_t sr = { madd3(wi, PT_EPS, hit_pos), wi };
        c->shadow_rays++;
        hit_t sh = traverse(s, sr, c);
        tap_shadow(c, dist);
        if (sh.t > 0.0f && sh.t < dist - PT_EPS * 2.0f) { ls.wi = wi; ls.pdf = 0.0f; return ls; }
        float att = 1.0f / (dist * dist);
        ls.intensity = scale3(scale3(ld3(lt->color), lt->intensity), att);
        ls.wi = wi;
        ls.pdf = inv_n * 10000.0f;
    } else if (lt->light_type == PTMI_LIGHT_EMISSIVE) {               /* :439-486 */
        ptmi_triangle zero_t; memset(&zero_t, 0, sizeof zero_t);
        const ptmi_triangle *T = lt->triangle_index < s->n_tris ? &s->tris[lt->triangle_index] : &zero_t;
        float r1 = rng_f(rng), r2 = rng_f(rng);
        float sq = sqrtf(r1);
        float u = 1.0f - sq;
        float v = r2 * sq;
        float w = 1.0f - u - v;
        v3 lp = lincomb3(ld3(T->v0), w, ld3(T->v1), u, ld3(T->v2), v);
        v3 n = normalize3(lincomb3(ld3(T->n0), w, ld3(T->n1), u, ld3(T->n2), v));
        v3 to_l = sub3(lp, hit_pos);
        float dist = length3(to_l);
        v3 wi = vdiv3(to_l, dist);
        ray_t sr = { madd3(wi, PT_EPS, hit_pos), wi };
        c->shadow_rays++;
        hit_t sh = traverse(s, sr, c);
        tap_shadow(c, dist);
        if (sh.t > 0.0f && sh.t < dist - PT_EPS * 2.0f) { ls.wi = wi; ls.pdf = 0.0f; return ls; }
        v3 e1 = sub3(ld3(T->v1), ld3(T->v0)), e2 = sub3(ld3(T->v2), ld3(T->v0));
        float area = length3(cross3(e1, e2)) * 0.5f;
        float cos_t = fabsf(dot3(n, neg3(wi)));
        ls.pdf = (inv_n * (1.0f / area)) * (dist * dist / max1(cos_t, PT_EPS));   /* :481 */
        ls.intensity = scale3(ld3(lt->color), lt->intensity);
        ls.wi = wi;
    }
    return ls;
}

/* ------------------------------------------------------------------------- */
/* trace — pt.wgsl:638-709                                                     */
/* ------------------------------------------------------------------------- */
static v3 trace(const pto_scene *s, uint32_t *rng, ray_t ray, uint32_t max_bounces, int do_mis,
                counters_t *c, float *log16, int *n_log) {
    v3 thr = V3(1.0f, 1.0f, 1.0f), res = V3(0.0f, 0.0f, 0.0f);
    ray_t cur = ray;
    int nl = 0;
    for (uint32_t bounce = 0; bounce < max_bounces; bounce++) {
        hit_t raw;
        c->segments++;
        hitinfo_t hit = scene_intersect(s, cur, c, &raw);                    /* :644 */
        if (log16) {
            float *L = log16 + 16 * nl++;
            L[0] = cur.o.x; L[1] = cur.o.y; L[2] = cur.o.z; L[3] = cur.d.x; L[4] = cur.d.y; L[5] = cur.d.z;
            L[6] = thr.x; L[7] = thr.y; L[8] = thr.z; L[9] = res.x; L[10] = res.y; L[11] = res.z;
            memcpy(&L[12], rng, 4); L[13] = raw.t; memcpy(&L[14], &raw.tri, 4); L[15] = 1.0f;
        }
        if (hit.t < 0.0f) {                                                  /* :646-649 */
            /* `result += throughput * vec3f(0.0)`: nothing for a finite throughput, NaN in every component whose
             * throughput is infinite or NaN (found by oracle/pt_literal.c, round 3; the kernels follow: shade.hip) */
            res = add3(res, mul3(thr, V3(0.0f, 0.0f, 0.0f)));
            break;
        }
        c->closest_hits++;
        if (hit.emission.x > 0.0f || hit.emission.y > 0.0f || hit.emission.z > 0.0f) {  /* :652 */
            float att = 1.0f / (1.0f + hit.t * hit.t);
            float k = hit.emissive_strength;
            res = V3(res.x + thr.x * hit.emission.x * k * att,
                     res.y + thr.y * hit.emission.y * k * att,
                     res.z + thr.z * hit.emission.z * k * att);              /* :656 */
            break;
        }
        /* :661. A scene without lights cannot be bound in the reference (WebGPU
         * rejects a zero-sized storage buffer); here it simply has no NEE. */
        if (do_mis && s->n_lights > 0u && hit.transmission == 0.0f && hit.is_front) {
            light_sample_t ls = sample_light(s, rng, hit.position, c);
            if (ls.pdf > 0.0f) {
                v3 Vv = neg3(normalize3(cur.d));
                v4 ev = eval_bsdf(&hit, hit.normal, Vv, ls.wi, hit.is_front);
                float wmis = power_heuristic(1.0f, ls.pdf, 1.0f, ev.w);
                v3 direct = vdiv3(scale3(mul3(ls.intensity, V3(ev.x, ev.y, ev.z)), wmis),
                                  max1(ls.pdf, PT_EPS));                     /* :674 */
                res = add3(res, mul3(thr, direct));                          /* :675 */
            }
        }
        v3 dir = sample_bsdf(rng, &hit, cur, hit.is_front);                  /* :680 */
        v4 ev = eval_bsdf(&hit, hit.normal, neg3(normalize3(cur.d)), dir, hit.is_front);
        if (ev.w <= 0.0f) break;                                             /* :685 (dead) */
        cur.o = madd3(dir, PT_EPS, hit.position);                            /* :691 */
        cur.d = normalize3(dir);                                             /* :692 */
        thr = mul3(thr, vdiv3(V3(ev.x, ev.y, ev.z), max1(ev.w, PT_EPS)));    /* :696 */
        if (bounce > 2u) {                                                   /* :699-705 */
            float p = max1(max1(thr.x, thr.y), thr.z);
            if (rng_f(rng) > p) break;
            thr = vdiv3(thr, p);
        }
    }
    if (log16) {
        float *L = log16 + 16 * nl++;
        L[0] = cur.o.x; L[1] = cur.o.y; L[2] = cur.o.z; L[3] = cur.d.x; L[4] = cur.d.y; L[5] = cur.d.z;
        L[6] = thr.x; L[7] = thr.y; L[8] = thr.z; L[9] = res.x; L[10] = res.y; L[11] = res.z;
        memcpy(&L[12], rng, 4); L[13] = 0.0f; L[14] = 0.0f; L[15] = 0.0f;
    }
    if (n_log) *n_log = nl;
    return res;
}

/* ------------------------------------------------------------------------- */
/* main — pt.wgsl:712-762                                                      */
/* ------------------------------------------------------------------------- */
static ray_t camera_ray(const ptmi_camera *cam, uint32_t x, uint32_t y, uint32_t frame, uint32_t *rng) {
    *rng = pto_seed(x, y, frame);                                            /* :719 */
    float jx = rng_f(rng), jy = rng_f(rng);
    float px = (float)x + jx, py = (float)y + jy;                            /* :723 */
    float uvx = (px / (float)cam->width) * 2.0f - 1.0f;                      /* :724 */
    float uvy = (py / (float)cam->height) * 2.0f - 1.0f;
    float th = tan1(cam->fov * 0.5f);
    v3 fw = ld3(cam->forward), rt = ld3(cam->right), up = ld3(cam->up), pos = ld3(cam->position);
    v3 a = scale3(scale3(scale3(rt, uvx), th), cam->aspect);                 /* :729 */
    v3 b = scale3(scale3(up, uvy), th);                                      /* :730 */
    v3 dir = normalize3(add3(add3(fw, a), b));
    v3 org = pos;
    if (cam->aperture > 0.0f) {                                              /* :736-748 */
        v3 focal = madd3(dir, cam->focus_distance, pos);
        float r = sqrtf(rng_f(rng)) * cam->aperture;
        float theta = rng_f(rng) * 2.0f * PT_PI;
        float st, ct; sincos1(theta, &st, &ct);
        v3 off = madd3(up, r * st, scale3(rt, r * ct));
        org = add3(pos, off);
        dir = normalize3(sub3(focal, org));
    }
    ray_t r = { org, dir };
    return r;
}

int pto_raygen(const ptmi_camera *cam, uint32_t n, const uint32_t *xs, const uint32_t *ys,
               const uint32_t *frames, float *o3, float *d3, uint32_t *rng_out) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t rng;
        ray_t r = camera_ray(cam, xs[i], ys[i], frames[i], &rng);
        o3[3 * i] = r.o.x; o3[3 * i + 1] = r.o.y; o3[3 * i + 2] = r.o.z;
        d3[3 * i] = r.d.x; d3[3 * i + 1] = r.d.y; d3[3 * i + 2] = r.d.z;
        if (rng_out) rng_out[i] = rng;
    }
    return 0;
}

static void add_counters(pto_stats *st, const counters_t *c) {
    st->segments += c->segments; st->shadow_rays += c->shadow_rays;
    st->nodes_visited += c->nodes_visited; st->tris_tested += c->tris_tested;
    st->closest_hits += c->closest_hits;
    if (c->max_stack > st->max_stack) st->max_stack = c->max_stack;
}

int pto_intersect(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                  float *t, uint32_t *tri, float *u, float *v, pto_stats *st) {
    counters_t tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel
    {
        counters_t c; memset(&c, 0, sizeof c);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            ray_t r = { ld3(o3 + 3 * i), ld3(d3 + 3 * i) };
            hit_t h = traverse(s, r, &c);
            t[i] = h.t; tri[i] = h.tri; u[i] = h.u; v[i] = h.v;
        }
#pragma omp critical
        {
            tot.nodes_visited += c.nodes_visited; tot.tris_tested += c.tris_tested;
            if (c.max_stack > tot.max_stack) tot.max_stack = c.max_stack;
        }
    }
    if (st) add_counters(st, &tot);
    return 0;
}

int pto_occluded(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                 const float *dist, uint8_t *occluded, pto_stats *st) {
    counters_t tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel
    {
        counters_t c; memset(&c, 0, sizeof c);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            ray_t r = { ld3(o3 + 3 * i), ld3(d3 + 3 * i) };
            hit_t h = traverse(s, r, &c);
            float dd = dist ? dist[i] : -1.0f;
            if (dd < 0.0f) occluded[i] = h.t > 0.0f;                          /* :394 */
            else occluded[i] = (h.t > 0.0f && h.t < dd - PT_EPS * 2.0f);      /* :423, :465 */
        }
#pragma omp critical
        { tot.nodes_visited += c.nodes_visited; tot.tris_tested += c.tris_tested; }
    }
    if (st) add_counters(st, &tot);
    return 0;
}

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

int pto_render(const pto_scene *s, const ptmi_camera *cam, uint32_t n_frames,
               const pto_options *opt, float *out, pto_stats *st) {
    uint32_t W = cam->width, H = cam->height;
    uint32_t y0 = opt ? opt->y0 : 0u, y1 = (opt && opt->y1) ? opt->y1 : H;
    uint32_t maxb = opt ? opt->max_bounces : 8u;
    int do_mis = opt ? (int)opt->do_mis : 1;
    if (y1 > H) y1 = H;
    if (y0 > y1) return -1;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = (opt && opt->threads) ? (int)opt->threads : omp_get_max_threads();
#endif
    counters_t tot; memset(&tot, 0, sizeof tot);
    double t0 = now_s();
#pragma omp parallel num_threads(nthreads)
    {
        counters_t c; memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = y0; y < (int64_t)y1; y++) {
            for (uint32_t x = 0; x < W; x++) {
                float *px = out + ((size_t)y * W + x) * 4;                   /* :753 */
                for (uint32_t k = 0; k < n_frames; k++) {
                    uint32_t frame = cam->frame_index + k;
                    uint32_t rng;
                    ray_t r = camera_ray(cam, x, (uint32_t)y, frame, &rng);
                    v3 col = trace(s, &rng, r, maxb, do_mis, &c, NULL, NULL);
                    col = V3(min1(col.x, 2.5f), min1(col.y, 2.5f), min1(col.z, 2.5f));  /* :751 */
                    if (frame > 0u) {                                        /* :754-759 */
                        float t = 1.0f / (float)(frame + 1u);
                        col = V3(mix1(px[0], col.x, t), mix1(px[1], col.y, t), mix1(px[2], col.z, t));
                    }
                    px[0] = col.x; px[1] = col.y; px[2] = col.z; px[3] = 0.0f;   /* :761 */
                }
            }
        }
#pragma omp critical
        {
            tot.segments += c.segments; tot.shadow_rays += c.shadow_rays;
            tot.nodes_visited += c.nodes_visited; tot.tris_tested += c.tris_tested;
            tot.closest_hits += c.closest_hits;
            if (c.max_stack > tot.max_stack) tot.max_stack = c.max_stack;
        }
    }
    if (st) {
        add_counters(st, &tot);
        st->paths += (uint64_t)(y1 - y0) * W * n_frames;
        st->seconds += now_s() - t0;
        st->threads = (uint32_t)nthreads;
    }
    return 0;
}

/* Every ray a render traces, with the traversal's result: rec9[9 i ..] = o.xyz, d.xyz, dist (0: closest-hit ray; < 0: shadow ray to
 * a directional light; > 0: shadow ray, the light's distance), t (-1: miss), tri (bits). Rows [y0, y1) x n_frames; at most max_rays
 * are stored (in no particular order); returns how many were traced. No image is written. */
uint64_t pto_render_tap(const pto_scene *s, const ptmi_camera *cam, uint32_t n_frames, const pto_options *opt,
                        float *rec9, uint64_t max_rays) {
    uint32_t W = cam->width, H = cam->height;
    uint32_t y0 = opt ? opt->y0 : 0u, y1 = (opt && opt->y1) ? opt->y1 : H;
    uint32_t maxb = opt ? opt->max_bounces : 8u;
    int do_mis = opt ? (int)opt->do_mis : 1;
    if (y1 > H) y1 = H;
    if (y0 > y1) return 0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = (opt && opt->threads) ? (int)opt->threads : omp_get_max_threads();
#endif
    uint64_t n = 0;
#pragma omp parallel num_threads(nthreads)
    {
        ray_tap_t tap = { rec9, max_rays, &n, 0 };
        counters_t c; memset(&c, 0, sizeof c);
        c.tap = &tap;
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = y0; y < (int64_t)y1; y++)
            for (uint32_t x = 0; x < W; x++)
                for (uint32_t k = 0; k < n_frames; k++) {
                    uint32_t rng;
                    ray_t r = camera_ray(cam, x, (uint32_t)y, cam->frame_index + k, &rng);
                    (void)trace(s, &rng, r, maxb, do_mis, &c, NULL, NULL);
                }
    }
    return n;
}

int pto_trace_path(const pto_scene *s, const ptmi_camera *cam, uint32_t x, uint32_t y,
                   uint32_t frame, const pto_options *opt, float *radiance3, float *log16) {
    counters_t c; memset(&c, 0, sizeof c);
    uint32_t rng; int nl = 0;
    ray_t r = camera_ray(cam, x, y, frame, &rng);
    v3 col = trace(s, &rng, r, opt ? opt->max_bounces : 8u, opt ? (int)opt->do_mis : 1, &c, log16, &nl);
    radiance3[0] = col.x; radiance3[1] = col.y; radiance3[2] = col.z;
    return nl;
}

/* Analysis aid (not a parity function): work counters of an ordered two-box descent with the
 * kernels' cull rule, per ray: out3[3i..] = wide-node steps, leaf visits, triangle tests. */
static void ordered_counts(const pto_scene *s, ray_t r, int cull, int pop_cull, float tlim_any, uint32_t *out3) {
    uint32_t steps = 0, leaves = 0, tris = 0;
    out3[0] = out3[1] = out3[2] = 0;
    if (s->n_nodes == 0) return;
    v3 inv = V3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    if (!box_test(&s->nodes[0], r.o, r.d, inv)) return;
    uint32_t stack[256]; float stackt[256]; int sp = 0;
    uint32_t cur = 0; float best = INFINITY; uint32_t btri = 0xFFFFFFFFu;
    int anyhit = tlim_any != 0.0f;                 /* tlim_any: 0 = closest hit, <0 directional, >0 distance limit */
    float limit = (anyhit && tlim_any > 0.0f && cull) ? __builtin_fmaf(tlim_any, 1.001f, 1e-4f) : INFINITY;
    for (;;) {
        const ptmi_bvh_node *n = &s->nodes[cur];
        if (n->triangle_count == 0) {
            steps++;
            const ptmi_bvh_node *L = &s->nodes[n->left], *R = &s->nodes[n->right];
            float tl, tr; int hl, hr;
            {   float t1x = (L->aabb_min[0] - r.o.x) * inv.x, t2x = (L->aabb_max[0] - r.o.x) * inv.x;
                float t1y = (L->aabb_min[1] - r.o.y) * inv.y, t2y = (L->aabb_max[1] - r.o.y) * inv.y;
                float t1z = (L->aabb_min[2] - r.o.z) * inv.z, t2z = (L->aabb_max[2] - r.o.z) * inv.z;
                tl = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
                float tm = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
                hl = tm >= tl && tm >= 0.0f; }
            {   float t1x = (R->aabb_min[0] - r.o.x) * inv.x, t2x = (R->aabb_max[0] - r.o.x) * inv.x;
                float t1y = (R->aabb_min[1] - r.o.y) * inv.y, t2y = (R->aabb_max[1] - r.o.y) * inv.y;
                float t1z = (R->aabb_min[2] - r.o.z) * inv.z, t2z = (R->aabb_max[2] - r.o.z) * inv.z;
                tr = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
                float tm = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
                hr = tm >= tr && tm >= 0.0f; }
            if (cull) { hl = hl && !(tl > limit); hr = hr && !(tr > limit); }
            if (hl && hr) {
                int lf = tl <= tr;
                stack[sp] = lf ? n->right : n->left; stackt[sp] = lf ? tr : tl; sp++;
                cur = lf ? n->left : n->right; continue;
            }
            if (hl) { cur = n->left; continue; }
            if (hr) { cur = n->right; continue; }
        } else {
            leaves++;
            int done = 0;
            for (uint32_t k = 0; k < n->triangle_count; k++) {
                uint32_t ti = n->triangle_offset + k; float u, v;
                tris++;
                float t = tri_test(&s->tris[ti], r.o, r.d, &u, &v);
                if (t > 0.0f) {
                    if (anyhit) { if (tlim_any < 0.0f || t < tlim_any) { done = 1; break; } }
                    else if (t < best || (t == best && ti < btri)) { best = t; btri = ti; if (cull) limit = __builtin_fmaf(t, 1.001f, 1e-4f); }
                }
            }
            if (done) break;
        }
        for (;;) {
            if (sp == 0) { out3[0] = steps; out3[1] = leaves; out3[2] = tris; return; }
            sp--; cur = stack[sp];
            if (!(pop_cull && cull && stackt[sp] > limit)) break;
        }
    }
    out3[0] = steps; out3[1] = leaves; out3[2] = tris;
}
void pto_ordered_counts(const pto_scene *s, uint32_t n, const float *o3, const float *d3, const float *tlim,
                        int cull, int pop_cull, uint32_t *out3) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        ray_t r = { ld3(o3 + 3 * i), ld3(d3 + 3 * i) };
        ordered_counts(s, r, cull, pop_cull, tlim ? tlim[i] : 0.0f, out3 + 3 * i);
    }
}

/* blit.wgsl:43-155 — tone map + gamma of the output buffer into a W x H canvas (row 0 = top). Plain libm
 * (log2f / powf): this pass is compared with a tolerance, not bit for bit. */
static float agx_contrast1(float v) {                                          /* blit.wgsl:54-65 */
    float x2 = v * v, x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * v + 31.96f * x4 - 6.868f * x2 * v + 0.4298f * x2 + 0.1191f * v - 0.00232f;
}
void pto_blit(const float *rgba, uint32_t W, uint32_t H, float *out_rgba) {
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    for (uint32_t j = 0; j < H; j++)
        for (uint32_t i = 0; i < W; i++) {
            float uvx = ((float)i + 0.5f) / (float)W, uvy = ((float)j + 0.5f) / (float)H;
            uint32_t x = f2u(uvx * (float)(W - 1u)), y = f2u((1.0f - uvy) * (float)(H - 1u));   /* :148-150 */
            const float *c = rgba + ((size_t)y * W + x) * 4;
            float cx = c[0] * 2.0f, cy = c[1] * 2.0f, cz = c[2] * 2.0f;                         /* :51, :134 */
            float r[3] = { 0.842479062253094f * cx + 0.0784335999999992f * cy + 0.0792237451477643f * cz,
                           0.0423282422610123f * cx + 0.878468636469772f * cy + 0.0791661274605434f * cz,
                           0.0423756549057051f * cx + 0.0784336f * cy + 0.879142973793104f * cz };
            for (int k = 0; k < 3; k++) {
                float l = min1(max1(log2f(r[k]), min_ev), max_ev);                              /* :81 */
                r[k] = agx_contrast1((l - min_ev) / (max_ev - min_ev));
            }
            float luma = r[0] * 0.2126f + r[1] * 0.7152f + r[2] * 0.0722f;                      /* :102-114 */
            for (int k = 0; k < 3; k++) r[k] = luma + (powf(r[k], 1.0f) - luma);
            float e[3] = { 1.19687900512017f * r[0] - 0.0980208811401368f * r[1] - 0.0990297440797205f * r[2],
                           -0.0528968517574562f * r[0] + 1.15190312990417f * r[1] - 0.0989611768448433f * r[2],
                           -0.0529716355144438f * r[0] - 0.0980434501171241f * r[1] + 1.15107367264116f * r[2] };
            float *o = out_rgba + ((size_t)j * W + i) * 4;
            for (int k = 0; k < 3; k++) o[k] = powf(powf(e[k], 2.2f), 1.0f / 2.2f);             /* :99, :46, :153 */
            o[3] = 1.0f;
        }
}

/* arithmetic-contract probe, same op codes as ptmi_debug_math (include/ptmi.h) */
void pto_math(int op, uint32_t n, const float *a, const float *b, const float *c, float *out) {
    for (uint32_t i = 0; i < n; i++) {
        float x = a[i], y = b ? b[i] : 0.0f, z = c ? c[i] : 0.0f, r = 0.0f, t;
        uint32_t bits;
        switch (op) {
        case 0: r = x / y; break;
        case 1: r = sqrtf(x); break;
        case 2: r = __builtin_fmaf(x, y, z); break;
        case 3: r = min1(x, y); break;
        case 4: r = max1(x, y); break;
        case 5: sincos1(x, &r, &t); break;
        case 6: sincos1(x, &t, &r); break;
        case 7: r = pow5(x); break;
        case 8: memcpy(&bits, &x, 4); r = (float)bits; break;
        case 9: bits = f2u(x); memcpy(&r, &bits, 4); break;
        case 10: r = x - truncf(x); break;
        case 11: r = tan1(x); break;
        default: break;
        }
        out[i] = r;
    }
}

/* ------------------------------------------------------------------------- */
/* probes for analytic KATs                                                   */
/* ------------------------------------------------------------------------- */
void pto_eval_bsdf(const float albedo[3], float roughness, float metallic, float transmission,
                   float ior, const float n[3], const float v[3], const float l[3], int front,
                   float out4[4]) {
    hitinfo_t h; memset(&h, 0, sizeof h);
    h.albedo = ld3(albedo); h.roughness = roughness; h.metallic = metallic;
    h.transmission = transmission; h.ior = ior; h.normal = ld3(n);
    v4 r = eval_bsdf(&h, ld3(n), ld3(v), ld3(l), front);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
float pto_distribution_ggx(const float n[3], const float h[3], float roughness) {
    return distribution_ggx(ld3(n), ld3(h), roughness);
}
float pto_power_heuristic(float nf, float fpdf, float ng, float gpdf) {
    return power_heuristic(nf, fpdf, ng, gpdf);
}
void pto_cosine_direction(uint32_t *state_io, float out3[3]) {
    v3 d = random_cosine_direction(state_io);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}
void pto_sample_ggx_normal(uint32_t *state_io, const float n[3], float roughness, float out3[3]) {
    v3 d = sample_ggx_normal(state_io, ld3(n), roughness);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}
