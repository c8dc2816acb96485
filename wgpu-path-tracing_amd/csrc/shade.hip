// shade.hip — the shade / next-event kernel of the wavefront path tracer.
//
// One lane per queued path segment. From the hit record (t, u, v, triangle) written by
// `extend` it rebuilds the shading state of the closest hit, applies emission, emits the
// next-event (shadow) record, samples and evaluates the BSDF, advances the ray and
// throughput, and plays Russian roulette — one iteration of the bounce loop of the
// reference (src/shader/pt.wgsl:638-709) minus its two traversals. All RNG draws of the
// bounce happen here, in the reference's order; the u32 RNG state
// travels in O.w. Survivors and emitted shadow records are flagged with one wave ballot
// each (64 paths -> one u64 word) for the ordered compaction kernel.
#include "pt_device.h"
#include "pt_math.h"

namespace {

struct HitInfo {                       // pt.wgsl:86-101 (fields the bounce loop reads)
    v3 position; float t;
    v3 normal;
    v3 albedo;
    float roughness, metallic, transmission, ior;
    v3 emission; float emissive_strength;
    bool is_front;
};

PT_DEV v3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

PT_DEV v4 atlas_load(const DevScene &sc, uint32_t x, uint32_t y) {
    v4 r; r.x = r.y = r.z = r.w = 0.0f;
    if (sc.atlas_fmt == 0u || x >= sc.atlas_w || y >= sc.atlas_h) return r;      // out of bounds reads zero
    size_t idx = ((size_t)y * sc.atlas_w + x);
    if (sc.atlas_fmt == 1u) {
        const uint2 raw = reinterpret_cast<const uint2 *>(sc.atlas)[idx];        // 4 x f16
        union { uint32_t u; _Float16 h[2]; } a, b;
        a.u = raw.x; b.u = raw.y;
        r.x = (float)a.h[0]; r.y = (float)a.h[1]; r.z = (float)b.h[0]; r.w = (float)b.h[1];
    } else {
        const float4 t = reinterpret_cast<const float4 *>(sc.atlas)[idx];
        r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    }
    return r;
}

// getTextureColor, pt.wgsl:112-120
PT_DEV v4 texture_color(const DevScene &sc, const ptmi_atlas_rect &tx, float uvx, float uvy, v4 fallback) {
    if (tx.w == 0u || tx.h == 0u) return fallback;
    float fx = uvx - __builtin_truncf(uvx);             // uv % 1.0 (exact)
    float fy = uvy - __builtin_truncf(uvy);
    float ax = (float)tx.x + fx * (float)tx.w;
    float ay = (float)tx.y + fy * (float)tx.h;
    return atlas_load(sc, f2u(ax), f2u(ay));
}

// rayTriangleIntersect, pt.wgsl:159-226, for the closest hit only. The hit record carries (t, triangle); the barycentric
// (u, v) are the ones `extend` computed when it accepted the hit — recomputed here by the same tri_test on the same
// operands (e1, e2 are the same single IEEE subtractions the traversal image was built with, ptmi_api.hip).
PT_DEV HitInfo make_hitinfo(const DevScene &sc, v3 ro, v3 rd, float t, uint32_t tri) {
    HitInfo hi;
    const ptmi_triangle &T = sc.tris[tri];
    // the whole triangle record is requested before the first use: the range tests of the short reciprocal / square root below
    // are branches, and a load placed after one is not issued before it
    const v3 tv0 = ld3(T.v0), tv1 = ld3(T.v1), tv2 = ld3(T.v2);
    const v3 tn0 = ld3(T.n0), tn1 = ld3(T.n1), tn2 = ld3(T.n2);
    const float u0x = T.uv0[0], u0y = T.uv0[1], u1x = T.uv1[0], u1y = T.uv1[1], u2x = T.uv2[0], u2y = T.uv2[1];
    const uint32_t mi = T.material_index;
    v3 v0 = tv0;
    v3 e1 = sub3(tv1, v0), e2 = sub3(tv2, v0);
    float u, v;
    (void)tri_test(v0, e1, e2, ro, rd, u, v);
    hi.t = t;
    hi.position = madd3(rd, t, ro);
    float w = 1.0f - u - v;
    v3 geo_n = normalize3(cross3(e1, e2));
    v3 n_i = normalize3(lincomb3(tn0, w, tn1, u, tn2, v));
    float uvx = fma1(u2x, v, fma1(u1x, u, u0x * w));
    float uvy = fma1(u2y, v, fma1(u1y, u, u0y * w));
    hi.is_front = dot3(geo_n, rd) < 0.0f;
    ptmi_material m;
    if (mi < sc.n_mats) m = sc.mats[mi];
    else __builtin_memset(&m, 0, sizeof m);
    v4 one; one.x = one.y = one.z = one.w = 1.0f;
    v4 alb = texture_color(sc, m.albedo_map, uvx, uvy, one);
    hi.albedo = mk3(alb.x * m.base_color[0], alb.y * m.base_color[1], alb.z * m.base_color[2]);
    v4 pbr = texture_color(sc, m.pbr_map, uvx, uvy, one);
    hi.metallic = pbr.z * m.metallic;
    hi.roughness = max1(pbr.y * m.roughness, 0.04f);
    hi.transmission = m.transmission;
    hi.ior = m.ior;
    v4 em = texture_color(sc, m.emissive_map, uvx, uvy, one);
    hi.emission = mk3(em.x * m.emission[0], em.y * m.emission[1], em.z * m.emission[2]);
    hi.emissive_strength = m.emissive_strength;
    v4 flat; flat.x = 0.5f; flat.y = 0.5f; flat.z = 1.0f; flat.w = 1.0f;
    v4 nm = texture_color(sc, m.normal_map, uvx, uvy, flat);
    if (nm.x != 0.5f || nm.y != 0.5f || nm.z != 1.0f) {
        float du1x = u1x - u0x, du1y = u1y - u0y;
        float du2x = u2x - u0x, du2y = u2y - u0y;
        float rr = rcp1(fma1(du1x, du2y, -(du1y * du2x)));
        v3 tg = mk3(fma1(e1.x, du2y, -(e2.x * du1y)) * rr, fma1(e1.y, du2y, -(e2.y * du1y)) * rr,
                    fma1(e1.z, du2y, -(e2.z * du1y)) * rr);
        tg = normalize3(tg);
        v3 N = n_i;
        v3 Tn = normalize3(madd3(N, -dot3(N, tg), tg));
        v3 Bn = normalize3(cross3(N, Tn));
        float tx = nm.x * 2.0f - 1.0f, ty = nm.y * 2.0f - 1.0f, tz = nm.z * 2.0f - 1.0f;
        hi.normal = normalize3(lincomb3(Tn, tx, Bn, ty, N, tz));
    } else {
        hi.normal = n_i;
    }
    return hi;
}

PT_DEV float distribution_ggx(v3 N, v3 H, float roughness) {   // pt.wgsl:316-325
    float a = roughness * roughness;
    float a2 = a * a;
    float ndh = max1(dot3(N, H), 0.0f);
    float ndh2 = ndh * ndh;
    float denom = ndh2 * (a2 - 1.0f) + 1.0f;
    return max1(a2 / (PT_PI * denom * denom), 0.0f);
}
PT_DEV float geometry_schlick_ggx(float ndv, float roughness) { // pt.wgsl:328-332
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return ndv / (ndv * (1.0f - k) + k);
}
PT_DEV float geometry_smith(v3 N, v3 V, v3 L, float roughness) { // pt.wgsl:334-340
    float ndv = max1(dot3(N, V), 0.0f);
    float ndl = max1(dot3(N, L), 0.0f);
    float g2 = geometry_schlick_ggx(ndv, roughness);
    float g1 = geometry_schlick_ggx(ndl, roughness);
    return g1 * g2;
}
PT_DEV v3 fresnel_schlick(float cos_theta, v3 F0) {             // pt.wgsl:343-345
    float p = pow5(1.0f - cos_theta);
    return mk3(fma1(1.0f - F0.x, p, F0.x), fma1(1.0f - F0.y, p, F0.y), fma1(1.0f - F0.z, p, F0.z));
}
PT_DEV float reflectance(float cos_theta, float eta) {          // pt.wgsl:616-620
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cos_theta);
}
PT_DEV void construct_tbn(v3 N, v3 &T, v3 &B) {                 // pt.wgsl:624-634
    T = mk3(1.0f, 0.0f, 0.0f);
    if (__builtin_fabsf(N.x) > 0.9f) T = mk3(0.0f, 1.0f, 0.0f);
    B = normalize3(cross3(N, T));
    T = normalize3(cross3(B, N));
}
// sampleGGXNormal, pt.wgsl:348-364, from its two draws' products: (sin, cos) of 2 pi r1, r2, and the frame constructTBN gives for `normal`
PT_DEV v3 ggx_normal_from(float sp, float cp, float r2, v3 normal, v3 T, v3 B, float roughness) {
    float a = roughness * roughness;
    float cos_t = sqrt1((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    float sin_t = sqrt1(1.0f - cos_t * cos_t);
    return normalize3(lincomb3(T, sin_t * cp, B, sin_t * sp, normal, cos_t));
}
PT_DEV float power_heuristic(float nf, float fpdf, float ng, float gpdf) { // pt.wgsl:492-496
    float f = nf * fpdf, g = ng * gpdf;
    return (f * f) / (f * f + g * g);
}
// sampleBSDF, pt.wgsl:498-546. The three lobes of the reference each start the same way — two draws, (sin, cos) of 2 pi r1, the frame
// constructTBN builds around the shading normal (randomCosineDirection + constructTBN :299-307 / :624-634 for the diffuse lobe,
// sampleGGXNormal :348-364 for the other two) — and a wave of bounce rays holds lanes of all three: written lobe by lobe it would
// execute that prefix three times and the GGX half-vector twice, each time for a part of its lanes. Here the common part runs once for
// the whole wave and the GGX half-vector once for the specular and the transmissive lanes together: the same operations on the same
// operands in the same order for every lane (bit-identical results, same RNG draws), a fifth fewer instructions for a mixed wave.
PT_DEV v3 sample_bsdf(uint32_t &rng, const HitInfo &h, v3 rd, bool front) {
    v3 V = neg3(normalize3(rd));
    float diffuse_p = (1.0f - h.metallic) * (1.0f - h.transmission);
    float specular_p = h.metallic;
    float r = rng_f(rng);
    const float r1 = rng_f(rng), r2 = rng_f(rng);
    const float phi = (2.0f * PT_PI) * r1;
    float sp, cp; sincos1(phi, sp, cp);
    v3 T, B; construct_tbn(h.normal, T, B);
    if (r < diffuse_p) {                                            // randomCosineDirection in the frame of the normal
        const float z = sqrt1(1.0f - r2), sr = sqrt1(r2);
        return lincomb3(T, cp * sr, B, sp * sr, h.normal, z);
    }
    float rough = max1(h.roughness, 0.04f);
    v3 N = ggx_normal_from(sp, cp, r2, h.normal, T, B, rough);
    if (r < diffuse_p + specular_p) return reflect3(neg3(V), N);
    float eta = front ? rcp1(h.ior) : h.ior;
    if (!front) N = neg3(N);
    float cos_t = dot3(N, V);
    float sin_t = sqrt1(1.0f - cos_t * cos_t);
    bool cannot_refract = eta * sin_t > 1.0f;
    float F = reflectance(__builtin_fabsf(cos_t), eta);
    if (cannot_refract || (rng_f(rng) < F)) return reflect3(neg3(V), N);   // short-circuit: draw only if needed
    return refract3(neg3(V), N, eta);
}
// evalBSDF, pt.wgsl:548-614: (f*cos, pdf)
PT_DEV v4 eval_bsdf(const HitInfo &h, v3 normal, v3 V, v3 L, bool front) {
    v3 H = normalize3(add3(V, L));
    float ndl = max1(dot3(normal, L), 0.0f);
    float ndv = max1(dot3(normal, V), 0.0f);
    float ndh = max1(dot3(normal, H), 0.0f);
    float vdh = max1(dot3(V, H), 0.0f);
    v3 F0 = mk3(mix1(0.04f, h.albedo.x, h.metallic), mix1(0.04f, h.albedo.y, h.metallic),
                mix1(0.04f, h.albedo.z, h.metallic));
    v3 F = fresnel_schlick(vdh, F0);
    float G = geometry_smith(normal, V, L, h.roughness);
    float D = distribution_ggx(normal, H, h.roughness);
    float one_m_tr = 1.0f - h.transmission;
    v3 kD = mk3((1.0f - F.x) * one_m_tr, (1.0f - F.y) * one_m_tr, (1.0f - F.z) * one_m_tr);
    v3 diffuse = vdiv3(mul3(kD, h.albedo), PT_PI);
    v3 specular = vdiv3(scale3(scale3(F, G), D), max1(4.0f * ndv * ndl, PT_EPS));
    v3 bsdf = mk3(0.0f, 0.0f, 0.0f);
    float pdf = 0.0f;
    if (h.transmission > 0.0f) {
        float eta = front ? rcp1(h.ior) : h.ior;
        float cos_t = dot3(normal, V);
        float Ft = reflectance(__builtin_fabsf(cos_t), eta);
        bsdf = scale3(h.albedo, 1.0f - Ft);
        pdf = (1.0f - h.metallic) * h.transmission;
    } else {
        bsdf = scale3(add3(diffuse, specular), ndl);
        float diffuse_p = (1.0f - h.metallic) * (1.0f - h.transmission);
        float specular_p = h.metallic;
        float diffuse_pdf = ndl / PT_PI;
        float specular_pdf = D * ndh / (4.0f * vdh);
        pdf = diffuse_p * diffuse_pdf + specular_p * specular_pdf;
    }
    v4 r; r.x = bsdf.x; r.y = bsdf.y; r.z = bsdf.z; r.w = max1(pdf, PT_EPS);
    return r;
}

struct LightSample { v3 intensity; v3 wi; float pdf; float dist; bool traced; };   // dist < 0: directional; traced: the
                                                                                  // reference shoots its shadow ray for this sample

// sampleLight, pt.wgsl:374-489, without its traversal: the occlusion test is the
// `shadow` kernel's; pdf = 0 means "no record" (the :413-415 early-out).
PT_DEV LightSample sample_light(const DevScene &sc, uint32_t &rng, v3 hit_pos) {
    LightSample ls;
    ls.intensity = mk3(0.0f, 0.0f, 0.0f); ls.wi = mk3(0.0f, 0.0f, 0.0f); ls.pdf = 0.0f; ls.dist = -1.0f; ls.traced = false;
    const uint32_t nl = sc.n_lights;
    const ptmi_light lt = sc.lights[rng_int(rng, 0u, nl - 1u)];
    const float inv_n = rcp1((float)nl);
    if (lt.light_type == PTMI_LIGHT_DIRECTIONAL) {
        ls.wi = normalize3(neg3(ld3(lt.position)));
        ls.intensity = scale3(ld3(lt.color), lt.intensity);
        ls.pdf = inv_n * 1000.0f;
        ls.dist = -1.0f;
        ls.traced = true;
    } else if (lt.light_type == PTMI_LIGHT_POINT) {
        v3 to_l = sub3(ld3(lt.position), hit_pos);
        float dist = length3(to_l);
        if (dist > 100.0f) return ls;
        ls.wi = vdiv3(to_l, dist);
        float att = rcp1(dist * dist);
        ls.intensity = scale3(scale3(ld3(lt.color), lt.intensity), att);
        ls.pdf = inv_n * 10000.0f;
        ls.dist = dist;
        ls.traced = true;
    } else if (lt.light_type == PTMI_LIGHT_EMISSIVE) {
        ptmi_triangle T;
        if (lt.triangle_index < sc.n_tris) T = sc.tris[lt.triangle_index];
        else __builtin_memset(&T, 0, sizeof T);
        float r1 = rng_f(rng), r2 = rng_f(rng);
        float sq = sqrt1(r1);
        float u = 1.0f - sq;
        float v = r2 * sq;
        float w = 1.0f - u - v;
        v3 lp = lincomb3(ld3(T.v0), w, ld3(T.v1), u, ld3(T.v2), v);
        v3 n = normalize3(lincomb3(ld3(T.n0), w, ld3(T.n1), u, ld3(T.n2), v));
        v3 to_l = sub3(lp, hit_pos);
        float dist = length3(to_l);
        v3 wi = vdiv3(to_l, dist);
        v3 e1 = sub3(ld3(T.v1), ld3(T.v0)), e2 = sub3(ld3(T.v2), ld3(T.v0));
        float area = length3(cross3(e1, e2)) * 0.5f;
        float cos_t = __builtin_fabsf(dot3(n, neg3(wi)));
        ls.pdf = (inv_n * rcp1(area)) * (dist * dist / max1(cos_t, PT_EPS));
        ls.intensity = scale3(ld3(lt.color), lt.intensity);
        ls.wi = wi;
        ls.dist = dist;
        ls.traced = true;
    }
    return ls;
}

constexpr int SBLOCK = 256;
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 0           /* > 0: waves per SIMD asked of the register allocator (94 VGPRs = 5 waves by itself) */
#endif
#if PT_SHADE_WAVES > 0
#define PT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(PT_SHADE_WAVES)))
#else
#define PT_SHADE_ATTR
#endif

__global__ __launch_bounds__(SBLOCK) PT_SHADE_ATTR void k_shade(DevScene sc, DevPaths P, const uint32_t *__restrict__ queue,
                                                  const uint32_t *__restrict__ count_ptr,
                                                  const float2 *__restrict__ hits, DevShadow S,
                                                  uint64_t *__restrict__ alive_mask,
                                                  uint64_t *__restrict__ shadow_mask, ShadeParams sp) {
    const uint32_t count = *count_ptr;
    uint32_t n_skipped = 0, n_emitted = 0;  // lane 0 of each wave: one atomic per wave at the end
    for (uint32_t base = blockIdx.x * SBLOCK; base < count; base += gridDim.x * SBLOCK) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false, shadow = false, skipped = false, emitted = false;
        if (i < count) {
            const uint32_t q = queue ? queue[i] : i;                         // where this ray's state is
            const float2 h2 = ld_stream(&hits[i]);
            if (!(h2.x < 0.0f)) {                                            // pt.wgsl:646: miss adds zero
                const float4 o4 = ld_stream(&P.O[q]), d4 = ld_stream(&P.D[q]);
                const uint32_t p = q;                                            // the path id: where its radiance is
                uint32_t rng = __float_as_uint(o4.w);
                const v3 ro = xyz(o4), rd = xyz(d4);
                v3 thr = mk3(1.0f, 1.0f, 1.0f);                                      // pt.wgsl:639; raygen stores no throughput
                if (sp.bounce != 0u) { const float2 c2 = ld_stream(&P.C[q]); thr = mk3(d4.w, c2.x, c2.y); }
                const HitInfo hit = make_hitinfo(sc, ro, rd, h2.x, __float_as_uint(h2.y));
                if (hit.emission.x > 0.0f || hit.emission.y > 0.0f || hit.emission.z > 0.0f) {   // pt.wgsl:652-658
                    float att = rcp1(1.0f + hit.t * hit.t);
                    float k = hit.emissive_strength;
                    const v3 e = mk3(thr.x * hit.emission.x * k * att, thr.y * hit.emission.y * k * att, thr.z * hit.emission.z * k * att);
                    if (sp.emit_records) {          // the path ends here: its last addition to L, made by `shadow` in bounce order
                        st_stream(&S.SO[i], make_float4(0.0f, 0.0f, 0.0f, -2.0f));
                        st_stream(&S.SD[i], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(p)));
                        S.SC[i] = rgb_sc{e.x, e.y, e.z};
                        shadow = true; emitted = true;
                    } else {
                        const rgb_sc l = P.ldL(p);
                        P.stL(p, l.x + e.x, l.y + e.y, l.z + e.z);
                    }
                } else {
                    if (sp.do_mis && sc.n_lights > 0u && hit.transmission == 0.0f && hit.is_front) {   // pt.wgsl:661
                        LightSample ls = sample_light(sc, rng, hit.position);
                        if (ls.pdf > 0.0f) {
                            v3 V = neg3(normalize3(rd));
                            v4 ev = eval_bsdf(hit, hit.normal, V, ls.wi, hit.is_front);
                            float wmis = power_heuristic(1.0f, ls.pdf, 1.0f, ev.w);
                            v3 direct = vdiv3(scale3(mul3(ls.intensity, mk3(ev.x, ev.y, ev.z)), wmis),
                                              max1(ls.pdf, PT_EPS));          // pt.wgsl:674
                            v3 contrib = mul3(thr, direct);                   // pt.wgsl:675, added by `shadow`
                            // A contribution of exactly zero (the light is behind the surface: NdotL = 0) leaves the
                            // radiance unchanged whatever the shadow ray finds (x + 0 = x), so that ray is counted
                            // in the statistics like the reference's traversal but neither recorded nor traced.
                            if ((contrib.x != 0.0f) | (contrib.y != 0.0f) | (contrib.z != 0.0f)) {
                                v3 so = madd3(ls.wi, PT_EPS, hit.position);
                                st_stream(&S.SO[i], make_float4(so.x, so.y, so.z, ls.dist));
                                st_stream(&S.SD[i], make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __uint_as_float(p)));
                                S.SC[i] = rgb_sc{contrib.x, contrib.y, contrib.z};
                                shadow = true;
                            } else {
                                skipped = true;
                            }
                        } else if (ls.traced) {
                            skipped = true;         // the reference traces this sample (pt.wgsl:392/421/463) and then drops it:
                        }                           // its pdf is not > 0 (underflow, 0 * inf); counted, nothing to add
                    }
                    v3 dir = sample_bsdf(rng, hit, rd, hit.is_front);         // pt.wgsl:680
                    v4 ev = eval_bsdf(hit, hit.normal, neg3(normalize3(rd)), dir, hit.is_front);
                    if (!(ev.w <= 0.0f)) {                                    // pt.wgsl:685
                        v3 no = madd3(dir, PT_EPS, hit.position);             // pt.wgsl:691
                        v3 nd = normalize3(dir);
                        thr = mul3(thr, vdiv3(mk3(ev.x, ev.y, ev.z), max1(ev.w, PT_EPS)));   // pt.wgsl:696
                        alive = true;
                        if (sp.bounce > 2u) {                                 // pt.wgsl:699-705
                            float pr = max1(max1(thr.x, thr.y), thr.z);
                            if (rng_f(rng) > pr) alive = false;
                            else thr = vdiv3(thr, pr);
                        }
                        if (alive && sp.bounce + 1u < sp.max_bounces) {
                            st_stream(&P.O[q], make_float4(no.x, no.y, no.z, __uint_as_float(rng)));
                            st_stream(&P.D[q], make_float4(nd.x, nd.y, nd.z, thr.x));
                            st_stream(&P.C[q], make_float2(thr.y, thr.z));
                        }
                    }
                }
            } else if (sp.bounce != 0u) {
                // pt.wgsl:646-648: a miss adds `throughput * vec3f(0.0)` — nothing while the throughput is finite (x + +-0 = x, and
                // the radiance is never -0), NaN in every component whose throughput is infinite or NaN (degenerate materials
                // only; the camera ray's throughput is 1). Such a path leaves a record like an emissive hit's.
                const float tx = reinterpret_cast<const float *>(&P.D[q])[3];
                const float2 c2 = ld_stream(&P.C[q]);
                if (!(__builtin_isfinite(tx) & __builtin_isfinite(c2.x) & __builtin_isfinite(c2.y))) {
                    const uint32_t p = q;
                    const v3 e = mk3(tx * 0.0f, c2.x * 0.0f, c2.y * 0.0f);
                    if (sp.emit_records) {
                        st_stream(&S.SO[i], make_float4(0.0f, 0.0f, 0.0f, -2.0f));
                        st_stream(&S.SD[i], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(p)));
                        S.SC[i] = rgb_sc{e.x, e.y, e.z};
                        shadow = true; emitted = true;
                    } else {
                        const rgb_sc l = P.ldL(p);
                        P.stL(p, l.x + e.x, l.y + e.y, l.z + e.z);
                    }
                }
            }
        }
        const uint64_t am = __ballot(alive), sm = __ballot(shadow), zm = __ballot(skipped), em = __ballot(emitted);
        if ((threadIdx.x & 63u) == 0u && i < count) {
            alive_mask[i >> 6] = am;
            shadow_mask[i >> 6] = sm;
            n_skipped += (uint32_t)__popcll(zm);
            n_emitted += (uint32_t)__popcll(em);
        }
    }
    if (n_skipped) atomicAdd(&sp.stats[1], (unsigned long long)n_skipped);
    if (n_emitted) atomicAdd(&sp.stats[3], (unsigned long long)n_emitted);
}

}  // namespace

// This file is built twice (Makefile): once under the arithmetic contract (pt_launch_shade, the default and the only
// build the parity tests compare with the oracle), once with PT_SHADE_FAST and the compiler's fast division / square root /
// contraction for ptmi_options.perf_mode = 1 (pt_launch_shade_fast): same source, same RNG draws, same control flow.
#ifdef PT_SHADE_FAST
#define PT_LAUNCH_SHADE pt_launch_shade_fast
#else
#define PT_LAUNCH_SHADE pt_launch_shade
#endif
void PT_LAUNCH_SHADE(hipStream_t s, int blocks, const DevScene &sc, DevPaths p, const uint32_t *queue,
                     const uint32_t *count, const float2 *hits, DevShadow sh, uint64_t *alive_mask,
                     uint64_t *shadow_mask, ShadeParams sp) {
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(SBLOCK), 0, s, sc, p, queue, count, hits, sh, alive_mask,
                       shadow_mask, sp);
}

#ifndef PT_SHADE_FAST
namespace {
__global__ void k_hit_uv(uint32_t n, DevScene sc, DevPaths P, const float2 *__restrict__ hits, float2 *__restrict__ uv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 h = hits[i];
    float u = 0.0f, v = 0.0f;
    if (!(h.x < 0.0f)) {
        const ptmi_triangle &T = sc.tris[__float_as_uint(h.y)];
        const v3 v0 = ld3(T.v0);
        (void)tri_test(v0, sub3(ld3(T.v1), v0), sub3(ld3(T.v2), v0), xyz(P.O[i]), xyz(P.D[i]), u, v);
    }
    uv[i] = make_float2(u, v);
}
}  // namespace
void pt_launch_hit_uv(hipStream_t s, uint32_t n, const DevScene &sc, DevPaths p, const float2 *hits, float2 *uv) {
    hipLaunchKernelGGL(k_hit_uv, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, p, hits, uv);
}
#endif
