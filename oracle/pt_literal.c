/*
 * pt_literal.c — the LITERAL oracle (TEST INFRASTRUCTURE; see pt_oracle.h for who may use it and for
 * the "parity unpinned" statement): the reference's compute shader restated in C in the reference's OWN
 * shape, one function per WGSL function, independent of pt_oracle.c.
 *
 * Why a second file. pt_oracle.c's PT_STRICT build is literal in its ARITHMETIC only: it shares the
 * contract build's control flow and data handling (a traversal that keeps (t, u, v, triangle) and rebuilds
 * the shading state once for the winner, a 1024-entry guarded stack, the RNG state passed by pointer).
 * This file shares nothing with it but the byte layouts of include/ptmi_layout.h:
 *   - rayTriangleIntersect returns the WHOLE HitInfo (material fetch, texture reads, tangent frame) for
 *     every accepted candidate, and traverseBVH copies it when the candidate is nearer, as pt.wgsl:123-226
 *     and :248-291 do;
 *   - sampleLight shoots its own shadow ray through sceneIntersect and reads `.t` of a full HitInfo
 *     (pt.wgsl:392/421/463);
 *   - the RNG state is a private variable of the invocation (random.wgsl:1), here a field of `Inv`;
 *   - every WGSL operator is ONE IEEE-754 binary32 operation in source order: `/` is a division, nothing is
 *     fused (-ffp-contract=off), sin/cos/tan/pow/sqrt are libm's, vector builtins are spelled out from
 *     their WGSL definitions (dot = x*x + y*y + z*z left to right, normalize = v / length(v), ...).
 * tests/test_oracle.py::test_literal_transcription_equals_the_strict_build requires the two literal
 * restatements to agree BIT FOR BIT (radiance, counters, RNG states, first hits); tests/test_gpu_strict.py
 * compares the HIP path with THIS library.
 *
 * Behaviour the WGSL leaves open, fixed here the way pt_oracle.c documents it:
 * min/max of NaN give the other operand and max(+0, -0) = +0; u32(f32) truncates, saturates, NaN -> 0;
 * out-of-range buffer / texture reads give zero; rand() == 1.0 would index lights[N] -> clamped (D-9); a
 * 65th push on the 64-entry stack (D-14, undefined in the reference) is dropped.
 *
 * Compile: oracle/Makefile (gcc -O2 -std=c11 -fopenmp -ffp-contract=off -fno-fast-math; no -mfma needed).
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* pt.wgsl:3-5 */
static const float PI = 3.14159265359f;
static const float EPSILON = 1e-6f;

typedef struct { float x, y; } Vec2;
typedef struct { float x, y, z; } Vec3;
typedef struct { float x, y, z, w; } Vec4;

/* ---- WGSL operators and builtins on vec3f, one rounding per scalar operation ---------------------------- */
static Vec3 vec3(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
static Vec3 splat(float s) { return vec3(s, s, s); }
static Vec3 from3(const float *p) { return vec3(p[0], p[1], p[2]); }
static Vec3 vadd(Vec3 a, Vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static Vec3 vsub(Vec3 a, Vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static Vec3 vmul(Vec3 a, Vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static Vec3 vdivv(Vec3 a, Vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static Vec3 vscale(Vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static Vec3 vdivs(Vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static Vec3 vneg(Vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static float w_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static Vec3 w_cross(Vec3 a, Vec3 b) {
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static float w_length(Vec3 a) { return sqrtf(w_dot(a, a)); }
static Vec3 w_normalize(Vec3 a) { return vdivs(a, w_length(a)); }
static float w_max(float a, float b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    if (a == b) return signbit(a) ? b : a;          /* only differs for the two zeros: the positive one */
    return a > b ? a : b;
}
static float w_min(float a, float b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    if (a == b) return signbit(a) ? a : b;          /* the negative zero */
    return a < b ? a : b;
}
static float w_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static Vec3 w_reflect(Vec3 e1, Vec3 e2) {           /* e1 - 2 * dot(e2, e1) * e2 */
    float k = 2.0f * w_dot(e2, e1);
    return vsub(e1, vscale(e2, k));
}
static Vec3 w_refract(Vec3 e1, Vec3 e2, float e3) { /* WGSL spec: k = 1 - e3^2 (1 - dot(e2,e1)^2); k < 0 -> 0 */
    float d = w_dot(e2, e1);
    float k = 1.0f - e3 * e3 * (1.0f - d * d);
    if (k < 0.0f) return splat(0.0f);
    return vsub(vscale(e1, e3), vscale(e2, e3 * d + sqrtf(k)));
}
static Vec3 mat3_mul(Vec3 c0, Vec3 c1, Vec3 c2, Vec3 v) {      /* mat3x3f(c0, c1, c2) * v */
    return vadd(vadd(vscale(c0, v.x), vscale(c1, v.y)), vscale(c2, v.z));
}
static uint32_t to_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static float from_half(uint16_t h) {
    int e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m + 1024), e - 25);
    return (h & 0x8000) ? -v : v;
}

/* ---- structs of pt.wgsl:80-101 ------------------------------------------------------------------------------ */
typedef struct { Vec3 origin, direction; } Ray;
typedef struct {
    Vec3 position; float t;
    Vec3 normal; uint32_t materialIndex;
    Vec3 albedo; float alpha, roughness, metallic, transmission, ior;
    Vec3 emission; float emissiveStrength;
    Vec2 uv; int isFront;
    uint32_t triangle; float u, v;        /* not in the reference: what pto_intersect reports */
} HitInfo;

/* one shader invocation: its private RNG state (random.wgsl:1), the bindings, and this file's counters */
typedef struct {
    uint32_t rngState;
    const pto_scene *scene;
    uint64_t segments, shadowRays, nodes, tris, closestHits;
    uint32_t maxStack;
} Inv;

/* ---- random.wgsl ----------------------------------------------------------------------------------------------- */
static void initRNG(Inv *inv, uint32_t px, uint32_t py, uint32_t frame) {      /* :3-5 */
    inv->rngState = px + py * 1000u + frame * 100000u;
}
static uint32_t rand_word(Inv *inv) {                                          /* :8-10 */
    inv->rngState = inv->rngState * 747796405u + 2891336453u;
    uint32_t result = ((inv->rngState >> ((inv->rngState >> 28u) + 4u)) ^ inv->rngState) * 277803737u;
    result = (result >> 22u) ^ result;
    return result;
}
static float rnd(Inv *inv) {                                                   /* :11: the literal 4294967295.0 is 2^32 as an f32 */
    return (float)rand_word(inv) / 4294967295.0f;
}
static uint32_t randInt(Inv *inv, uint32_t lo, uint32_t hi) {                  /* :14-16 */
    uint32_t k = to_u32(rnd(inv) * (float)(hi - lo + 1u));
    if (k > hi - lo) k = hi - lo;                                              /* D-9 */
    return lo + k;
}

/* ---- pt.wgsl:112-120 ------------------------------------------------------------------------------------------- */
static Vec4 textureLoad(const pto_scene *s, uint32_t x, uint32_t y) {
    Vec4 r = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (!s->atlas || s->atlas_fmt == PTO_ATLAS_NONE || x >= s->atlas_w || y >= s->atlas_h) return r;
    size_t at = ((size_t)y * s->atlas_w + x) * 4u;
    if (s->atlas_fmt == PTO_ATLAS_RGBA16F) {
        const uint16_t *p = (const uint16_t *)s->atlas + at;
        r.x = from_half(p[0]); r.y = from_half(p[1]); r.z = from_half(p[2]); r.w = from_half(p[3]);
    } else {
        const float *p = (const float *)s->atlas + at;
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    }
    return r;
}
static float fmod1(float x) { return x - 1.0f * truncf(x / 1.0f); }            /* WGSL x % y = x - y * trunc(x / y) */
static Vec4 getTextureColor(const pto_scene *s, ptmi_atlas_rect texture, Vec2 uv, Vec4 fallback) {
    float ax = (float)texture.x + fmod1(uv.x) * (float)texture.w;
    float ay = (float)texture.y + fmod1(uv.y) * (float)texture.h;
    if (texture.w == 0u || texture.h == 0u) return fallback;                   /* select(load, fallback, cond) */
    return textureLoad(s, to_u32(ax), to_u32(ay));
}

/* ---- rayTriangleIntersect, pt.wgsl:123-226 ------------------------------------------------------------------- */
static HitInfo rayTriangleIntersect(Inv *inv, Ray ray, const ptmi_triangle *triangle) {
    const pto_scene *s = inv->scene;
    HitInfo hit;
    memset(&hit, 0, sizeof hit);
    hit.t = -1.0f;
    Vec3 v0 = from3(triangle->v0), v1 = from3(triangle->v1), v2 = from3(triangle->v2);
    Vec3 edge1 = vsub(v1, v0);
    Vec3 edge2 = vsub(v2, v0);
    Vec3 h = w_cross(ray.direction, edge2);
    float a = w_dot(edge1, h);
    if (fabsf(a) < EPSILON) return hit;
    float f = 1.0f / a;
    Vec3 sv = vsub(ray.origin, v0);
    float u = f * w_dot(sv, h);
    if (u < 0.0f || u > 1.0f) return hit;
    Vec3 q = w_cross(sv, edge1);
    float v = f * w_dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return hit;
    float t = f * w_dot(edge2, q);
    if (t > EPSILON) {
        hit.t = t;
        hit.u = u; hit.v = v;
        hit.position = vadd(ray.origin, vscale(ray.direction, t));
        float w = 1.0f - u - v;
        Vec3 geometryNormal = w_normalize(w_cross(edge1, edge2));
        Vec3 n0 = from3(triangle->n0), n1 = from3(triangle->n1), n2 = from3(triangle->n2);
        Vec3 interpolatedNormal = w_normalize(vadd(vadd(vscale(n0, w), vscale(n1, u)), vscale(n2, v)));
        /* tangent frame, :176-189 */
        Vec3 deltaPos1 = vsub(v1, v0), deltaPos2 = vsub(v2, v0);
        Vec2 deltaUV1 = { triangle->uv1[0] - triangle->uv0[0], triangle->uv1[1] - triangle->uv0[1] };
        Vec2 deltaUV2 = { triangle->uv2[0] - triangle->uv0[0], triangle->uv2[1] - triangle->uv0[1] };
        float r = 1.0f / (deltaUV1.x * deltaUV2.y - deltaUV1.y * deltaUV2.x);
        Vec3 tangent = w_normalize(vscale(vsub(vscale(deltaPos1, deltaUV2.y), vscale(deltaPos2, deltaUV1.y)), r));
        /* (the bitangent of :183 is computed and never read) */
        Vec3 N = interpolatedNormal;
        Vec3 T = w_normalize(vsub(tangent, vscale(N, w_dot(N, tangent))));
        Vec3 B = w_normalize(w_cross(N, T));
        hit.uv.x = triangle->uv0[0] * w + triangle->uv1[0] * u + triangle->uv2[0] * v;
        hit.uv.y = triangle->uv0[1] * w + triangle->uv1[1] * u + triangle->uv2[1] * v;
        hit.materialIndex = triangle->material_index;
        hit.isFront = w_dot(geometryNormal, ray.direction) < 0.0f;
        ptmi_material material;
        if (hit.materialIndex < s->n_mats) material = s->mats[hit.materialIndex];
        else memset(&material, 0, sizeof material);
        Vec4 ones = { 1.0f, 1.0f, 1.0f, 1.0f };
        Vec4 albedoValue = getTextureColor(s, material.albedo_map, hit.uv, ones);
        hit.albedo = vmul(vec3(albedoValue.x, albedoValue.y, albedoValue.z), from3(material.base_color));
        hit.alpha = albedoValue.w;
        Vec4 pbrValue = getTextureColor(s, material.pbr_map, hit.uv, ones);
        hit.metallic = pbrValue.z * material.metallic;
        hit.roughness = w_max(pbrValue.y * material.roughness, 0.04f);
        hit.transmission = material.transmission;
        hit.ior = material.ior;
        Vec4 emissiveValue = getTextureColor(s, material.emissive_map, hit.uv, ones);
        hit.emission = vmul(vec3(emissiveValue.x, emissiveValue.y, emissiveValue.z), from3(material.emission));
        hit.emissiveStrength = material.emissive_strength;
        Vec4 flat = { 0.5f, 0.5f, 1.0f, 1.0f };
        Vec4 nm4 = getTextureColor(s, material.normal_map, hit.uv, flat);
        Vec3 normalMap = vec3(nm4.x, nm4.y, nm4.z);
        if (normalMap.x != 0.5f || normalMap.y != 0.5f || normalMap.z != 1.0f) {
            Vec3 tangentNormal = vsub(vscale(normalMap, 2.0f), splat(1.0f));
            hit.normal = w_normalize(mat3_mul(T, B, N, tangentNormal));
        } else {
            hit.normal = interpolatedNormal;
        }
    }
    return hit;
}

/* ---- rayAABBIntersect, pt.wgsl:229-245 ------------------------------------------------------------------------ */
static int rayAABBIntersect(Ray ray, const ptmi_bvh_node *node) {
    Vec3 t1 = vdivv(vsub(from3(node->aabb_min), ray.origin), ray.direction);
    Vec3 t2 = vdivv(vsub(from3(node->aabb_max), ray.origin), ray.direction);
    Vec3 tmin = vec3(w_min(t1.x, t2.x), w_min(t1.y, t2.y), w_min(t1.z, t2.z));
    Vec3 tmax = vec3(w_max(t1.x, t2.x), w_max(t1.y, t2.y), w_max(t1.z, t2.z));
    float t_min = w_max(w_max(tmin.x, tmin.y), tmin.z);
    float t_max = w_min(w_min(tmax.x, tmax.y), tmax.z);
    return t_max >= t_min && t_max >= 0.0f;
}

/* ---- traverseBVH / sceneIntersect, pt.wgsl:248-296 ------------------------------------------------------------ */
static HitInfo traverseBVH(Inv *inv, Ray ray) {
    const pto_scene *s = inv->scene;
    uint32_t stack[64];
    uint32_t stackPtr = 0u;
    HitInfo closest;
    memset(&closest, 0, sizeof closest);
    closest.t = -1.0f;
    closest.triangle = 0xFFFFFFFFu;
    int hasHit = 0;
    if (s->n_nodes == 0u) return closest;          /* (an empty node buffer cannot be bound in the reference) */
    stack[stackPtr] = 0u;
    stackPtr += 1u;
    while (stackPtr > 0u) {
        stackPtr -= 1u;
        uint32_t nodeIdx = stack[stackPtr];
        if (nodeIdx >= s->n_nodes) continue;
        const ptmi_bvh_node *node = &s->nodes[nodeIdx];
        inv->nodes++;
        if (!rayAABBIntersect(ray, node)) continue;
        if (node->triangle_count > 0u) {
            for (uint32_t i = 0u; i < node->triangle_count; i++) {
                uint32_t triIdx = node->triangle_offset + i;
                if (triIdx >= s->n_tris) continue;
                inv->tris++;
                HitInfo hit = rayTriangleIntersect(inv, ray, &s->tris[triIdx]);
                if (hit.t > 0.0f && (hit.t < closest.t || !hasHit)) {
                    closest = hit;
                    closest.triangle = triIdx;
                    hasHit = 1;
                }
            }
        } else if (stackPtr + 2u <= 64u) {
            stack[stackPtr] = node->right;
            stackPtr += 1u;
            stack[stackPtr] = node->left;
            stackPtr += 1u;
            if (stackPtr > inv->maxStack) inv->maxStack = stackPtr;
        }
    }
    return closest;
}
static HitInfo sceneIntersect(Inv *inv, Ray ray) { return traverseBVH(inv, ray); }

/* ---- pt.wgsl:299-364 ---------------------------------------------------------------------------------------------- */
static Vec3 randomCosineDirection(Inv *inv) {
    float r1 = rnd(inv);
    float r2 = rnd(inv);
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * PI * r1;
    float x = cosf(phi) * sqrtf(r2);
    float y = sinf(phi) * sqrtf(r2);
    return vec3(x, y, z);
}
static float distributionGGX(Vec3 N, Vec3 H, float roughness) {
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH = w_max(w_dot(N, H), 0.0f);
    float NdotH2 = NdotH * NdotH;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    return w_max(a2 / (PI * denom * denom), 0.0f);
}
static float geometrySchlickGGX(float NdotV, float roughness) {
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}
static float geometrySmith(Vec3 N, Vec3 V, Vec3 L, float roughness) {
    float NdotV = w_max(w_dot(N, V), 0.0f);
    float NdotL = w_max(w_dot(N, L), 0.0f);
    float ggx2 = geometrySchlickGGX(NdotV, roughness);
    float ggx1 = geometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}
static Vec3 fresnelSchlick(float cosTheta, Vec3 F0) {
    float p = powf(1.0f - cosTheta, 5.0f);
    return vadd(F0, vscale(vsub(splat(1.0f), F0), p));
}
/* constructTBN, :624-634: the columns T, B (the third is N itself) */
static void constructTBN(Vec3 N, Vec3 *Tout, Vec3 *Bout) {
    Vec3 T = vec3(1.0f, 0.0f, 0.0f);
    if (fabsf(N.x) > 0.9f) T = vec3(0.0f, 1.0f, 0.0f);
    Vec3 B = w_normalize(w_cross(N, T));
    T = w_normalize(w_cross(B, N));
    *Tout = T; *Bout = B;
}
static Vec3 sampleGGXNormal(Inv *inv, Vec3 normal, float roughness) {
    float r1 = rnd(inv);
    float r2 = rnd(inv);
    float a = roughness * roughness;
    float phi = 2.0f * PI * r1;
    float cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    Vec3 N = vec3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    Vec3 T, B;
    constructTBN(normal, &T, &B);
    return w_normalize(mat3_mul(T, B, normal, N));
}

/* ---- sampleLight, pt.wgsl:366-489 ------------------------------------------------------------------------------ */
typedef struct { Vec3 intensity; uint32_t lightType; Vec3 wi; float pdf; } LightSample;

static LightSample sampleLight(Inv *inv, Vec3 hitPosition) {
    const pto_scene *s = inv->scene;
    uint32_t nLights = s->n_lights;
    ptmi_light light = s->lights[randInt(inv, 0u, nLights - 1u)];
    LightSample sample;
    sample.lightType = light.light_type;
    sample.intensity = splat(0.0f);
    sample.wi = splat(0.0f);
    sample.pdf = 0.0f;
    if (light.light_type == PTMI_LIGHT_DIRECTIONAL) {
        Vec3 wi = w_normalize(vneg(from3(light.position)));
        Ray shadowRay = { vadd(hitPosition, vscale(wi, EPSILON)), wi };
        inv->shadowRays++;
        HitInfo shadowHit = sceneIntersect(inv, shadowRay);
        if (shadowHit.t > 0.0f) {
            sample.intensity = splat(0.0f); sample.wi = wi; sample.pdf = 0.0f;
            return sample;
        }
        sample.intensity = vscale(from3(light.color), light.intensity);
        sample.wi = wi;
        sample.pdf = 1.0f / (float)nLights * 1000.0f;
    } else if (light.light_type == PTMI_LIGHT_POINT) {
        Vec3 toLight = vsub(from3(light.position), hitPosition);
        float dist = w_length(toLight);
        if (dist > 100.0f) return sample;
        Vec3 wi = vdivs(toLight, dist);
        Ray shadowRay = { vadd(hitPosition, vscale(wi, EPSILON)), wi };
        inv->shadowRays++;
        HitInfo shadowHit = sceneIntersect(inv, shadowRay);
        if (shadowHit.t > 0.0f && shadowHit.t < dist - EPSILON * 2.0f) {
            sample.intensity = splat(0.0f); sample.wi = wi; sample.pdf = 0.0f;
            return sample;
        }
        float attenuation = 1.0f / (dist * dist);
        sample.intensity = vscale(vscale(from3(light.color), light.intensity), attenuation);
        sample.wi = wi;
        sample.pdf = 1.0f / (float)nLights * 10000.0f;
    } else if (light.light_type == PTMI_LIGHT_EMISSIVE) {
        ptmi_triangle triangle;
        if (light.triangle_index < s->n_tris) triangle = s->tris[light.triangle_index];
        else memset(&triangle, 0, sizeof triangle);
        float r1 = rnd(inv);
        float r2 = rnd(inv);
        float u = 1.0f - sqrtf(r1);
        float v = r2 * sqrtf(r1);
        float w = 1.0f - u - v;
        Vec3 v0 = from3(triangle.v0), v1 = from3(triangle.v1), v2 = from3(triangle.v2);
        Vec3 lightPos = vadd(vadd(vscale(v0, w), vscale(v1, u)), vscale(v2, v));
        Vec3 normal = w_normalize(vadd(vadd(vscale(from3(triangle.n0), w), vscale(from3(triangle.n1), u)),
                                       vscale(from3(triangle.n2), v)));
        Vec3 toLight = vsub(lightPos, hitPosition);
        float dist = w_length(toLight);
        Vec3 wi = vdivs(toLight, dist);
        Ray shadowRay = { vadd(hitPosition, vscale(wi, EPSILON)), wi };
        inv->shadowRays++;
        HitInfo shadowHit = sceneIntersect(inv, shadowRay);
        if (shadowHit.t > 0.0f && shadowHit.t < dist - EPSILON * 2.0f) {
            sample.intensity = splat(0.0f); sample.wi = wi; sample.pdf = 0.0f;
            return sample;
        }
        Vec3 edge1 = vsub(v1, v0);
        Vec3 edge2 = vsub(v2, v0);
        float triangleArea = w_length(w_cross(edge1, edge2)) * 0.5f;
        float cosTheta = fabsf(w_dot(normal, vneg(wi)));
        sample.pdf = (1.0f / (float)nLights) * (1.0f / triangleArea) * (dist * dist / w_max(cosTheta, EPSILON));
        sample.intensity = vscale(from3(light.color), light.intensity);
        sample.wi = wi;
    }
    return sample;
}

/* ---- pt.wgsl:492-634 ---------------------------------------------------------------------------------------------- */
static float powerHeuristic(float nf, float fPdf, float ng, float gPdf) {
    float f = nf * fPdf;
    float g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
static float reflectance(float cosTheta, float eta) {
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosTheta, 5.0f);
}
static Vec3 sampleBSDF(Inv *inv, const HitInfo *hitInfo, Ray currentRay, int front) {
    Vec3 V = vneg(w_normalize(currentRay.direction));
    float diffuseProb = (1.0f - hitInfo->metallic) * (1.0f - hitInfo->transmission);
    float specularProb = hitInfo->metallic;
    float r = rnd(inv);
    if (r < diffuseProb) {
        Vec3 localDir = randomCosineDirection(inv);
        Vec3 T, B;
        constructTBN(hitInfo->normal, &T, &B);
        return mat3_mul(T, B, hitInfo->normal, localDir);
    } else if (r < diffuseProb + specularProb) {
        float roughness = w_max(hitInfo->roughness, 0.04f);
        Vec3 N = sampleGGXNormal(inv, hitInfo->normal, roughness);
        return w_reflect(vneg(V), N);
    } else {
        float eta = front ? 1.0f / hitInfo->ior : hitInfo->ior;
        float roughness = w_max(hitInfo->roughness, 0.04f);
        Vec3 N = sampleGGXNormal(inv, hitInfo->normal, roughness);
        if (!front) N = vneg(N);
        float cosTheta = w_dot(N, V);
        float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        int cannotRefract = eta * sinTheta > 1.0f;
        float F = reflectance(fabsf(cosTheta), eta);
        if (cannotRefract || (rnd(inv) < F)) return w_reflect(vneg(V), N);      /* || short-circuits: a draw only if needed */
        return w_refract(vneg(V), N, eta);
    }
}
static Vec4 evalBSDF(const HitInfo *hitInfo, Vec3 normal, Vec3 V, Vec3 L, int front) {
    Vec3 H = w_normalize(vadd(V, L));
    float NdotL = w_max(w_dot(normal, L), 0.0f);
    float NdotV = w_max(w_dot(normal, V), 0.0f);
    float NdotH = w_max(w_dot(normal, H), 0.0f);
    float VdotH = w_max(w_dot(V, H), 0.0f);
    Vec3 F0 = vec3(w_mix(0.04f, hitInfo->albedo.x, hitInfo->metallic), w_mix(0.04f, hitInfo->albedo.y, hitInfo->metallic),
                   w_mix(0.04f, hitInfo->albedo.z, hitInfo->metallic));
    Vec3 F = fresnelSchlick(VdotH, F0);
    float G = geometrySmith(normal, V, L, hitInfo->roughness);
    float D = distributionGGX(normal, H, hitInfo->roughness);
    Vec3 kD = vscale(vsub(splat(1.0f), F), 1.0f - hitInfo->transmission);
    Vec3 diffuse = vdivs(vmul(kD, hitInfo->albedo), PI);
    Vec3 specular = vdivs(vscale(vscale(F, G), D), w_max(4.0f * NdotV * NdotL, EPSILON));
    Vec3 bsdf = splat(0.0f);
    float pdf = 0.0f;
    if (hitInfo->transmission > 0.0f) {
        float eta = front ? 1.0f / hitInfo->ior : hitInfo->ior;
        float cosTheta = w_dot(normal, V);
        float F_transmission = reflectance(fabsf(cosTheta), eta);
        bsdf = vscale(hitInfo->albedo, 1.0f - F_transmission);      /* both arms of :585-593 are the same */
        pdf = (1.0f - hitInfo->metallic) * hitInfo->transmission;
    } else {
        bsdf = vscale(vadd(diffuse, specular), NdotL);
        float diffuseProb = (1.0f - hitInfo->metallic) * (1.0f - hitInfo->transmission);
        float specularProb = hitInfo->metallic;
        float diffusePdf = NdotL / PI;
        float specularPdf = D * NdotH / (4.0f * VdotH);
        pdf = diffuseProb * diffusePdf + specularProb * specularPdf;
    }
    Vec4 r = { bsdf.x, bsdf.y, bsdf.z, w_max(pdf, EPSILON) };
    return r;
}

/* ---- trace, pt.wgsl:638-709 --------------------------------------------------------------------------------------- */
static Vec3 trace(Inv *inv, Ray ray, uint32_t maxBounces, int doMis) {
    Vec3 throughput = splat(1.0f);
    Vec3 result = splat(0.0f);
    Ray currentRay = ray;
    for (uint32_t bounce = 0u; bounce < maxBounces; bounce++) {
        inv->segments++;
        HitInfo hit = sceneIntersect(inv, currentRay);
        if (hit.t < 0.0f) {
            result = vadd(result, vmul(throughput, splat(0.0f)));
            break;
        }
        inv->closestHits++;
        if (hit.emission.x > 0.0f || hit.emission.y > 0.0f || hit.emission.z > 0.0f) {
            float distance = hit.t;
            float attenuation = 1.0f / (1.0f + distance * distance);
            result = vadd(result, vscale(vscale(vmul(throughput, hit.emission), hit.emissiveStrength), attenuation));
            break;
        }
        if (doMis && inv->scene->n_lights > 0u && hit.transmission == 0.0f && hit.isFront) {
            LightSample lightSample = sampleLight(inv, hit.position);
            if (lightSample.pdf > 0.0f) {
                Vec3 V = vneg(w_normalize(currentRay.direction));
                Vec4 evalResult = evalBSDF(&hit, hit.normal, V, lightSample.wi, hit.isFront);
                Vec3 bsdfValue = vec3(evalResult.x, evalResult.y, evalResult.z);
                float bsdfPdf = evalResult.w;
                float misWeight = powerHeuristic(1.0f, lightSample.pdf, 1.0f, bsdfPdf);
                Vec3 directLight = vdivs(vscale(vmul(lightSample.intensity, bsdfValue), misWeight), w_max(lightSample.pdf, EPSILON));
                result = vadd(result, vmul(throughput, directLight));
            }
        }
        Vec3 bsdfDir = sampleBSDF(inv, &hit, currentRay, hit.isFront);
        Vec4 evalResult = evalBSDF(&hit, hit.normal, vneg(w_normalize(currentRay.direction)), bsdfDir, hit.isFront);
        Vec3 bsdfValue = vec3(evalResult.x, evalResult.y, evalResult.z);
        float bsdfPdf = evalResult.w;
        if (bsdfPdf <= 0.0f) break;
        currentRay.origin = vadd(hit.position, vscale(bsdfDir, EPSILON));
        currentRay.direction = w_normalize(bsdfDir);
        throughput = vmul(throughput, vdivs(bsdfValue, w_max(bsdfPdf, EPSILON)));
        if (bounce > 2u) {
            float p = w_max(w_max(throughput.x, throughput.y), throughput.z);
            if (rnd(inv) > p) break;
            throughput = vdivs(throughput, p);
        }
    }
    return result;
}

/* ---- main, pt.wgsl:712-762: the ray of pixel (x, y) at `frame`; leaves the RNG state behind it in inv ------------ */
static Ray cameraRay(Inv *inv, const ptmi_camera *camera, uint32_t px, uint32_t py, uint32_t frame) {
    initRNG(inv, px, py, frame);
    float pixelX = (float)px + rnd(inv);
    float pixelY = (float)py + rnd(inv);
    float uvx = (pixelX / (float)camera->width) * 2.0f - 1.0f;
    float uvy = (pixelY / (float)camera->height) * 2.0f - 1.0f;
    Vec3 forward = from3(camera->forward), right = from3(camera->right), up = from3(camera->up);
    Vec3 position = from3(camera->position);
    Vec3 rayDir = w_normalize(vadd(vadd(forward, vscale(vscale(vscale(right, uvx), tanf(camera->fov * 0.5f)), camera->aspect)),
                                   vscale(vscale(up, uvy), tanf(camera->fov * 0.5f))));
    Vec3 rayOrigin = position;
    if (camera->aperture > 0.0f) {
        Vec3 focalPoint = vadd(position, vscale(rayDir, camera->focus_distance));
        float r = sqrtf(rnd(inv)) * camera->aperture;
        float theta = rnd(inv) * 2.0f * PI;
        Vec3 offset = vadd(vscale(right, r * cosf(theta)), vscale(up, r * sinf(theta)));
        rayOrigin = vadd(position, offset);
        rayDir = w_normalize(vsub(focalPoint, rayOrigin));
    }
    Ray ray = { rayOrigin, rayDir };
    return ray;
}

/* ---- the entry points tests use on a literal oracle (same signatures as pt_oracle.c's) ------------------------ */
int pto_is_strict(void) { return 2; }                  /* 0: contract build, 1: pt_oracle.c's PT_STRICT build, 2: this file */

uint32_t pto_seed(uint32_t x, uint32_t y, uint32_t frame) { Inv inv; initRNG(&inv, x, y, frame); return inv.rngState; }
void pto_rand(uint32_t *state_io, uint32_t n, uint32_t *states, uint32_t *words, float *vals) {
    Inv inv; inv.rngState = *state_io;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t before = inv.rngState;
        float v = rnd(&inv);
        uint32_t after = inv.rngState;
        if (states) states[i] = after;
        if (vals) vals[i] = v;
        if (words) { inv.rngState = before; words[i] = rand_word(&inv); }
    }
    *state_io = inv.rngState;
}
uint32_t pto_rand_int(uint32_t *state_io, uint32_t lo, uint32_t hi) {
    Inv inv; inv.rngState = *state_io;
    uint32_t k = randInt(&inv, lo, hi);
    *state_io = inv.rngState;
    return k;
}
void pto_sincos(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
float pto_distribution_ggx(const float n[3], const float h[3], float roughness) { return distributionGGX(from3(n), from3(h), roughness); }
float pto_power_heuristic(float nf, float fpdf, float ng, float gpdf) { return powerHeuristic(nf, fpdf, ng, gpdf); }
void pto_eval_bsdf(const float albedo[3], float roughness, float metallic, float transmission, float ior,
                   const float n[3], const float v[3], const float l[3], int front, float out4[4]) {
    HitInfo h; memset(&h, 0, sizeof h);
    h.albedo = from3(albedo); h.roughness = roughness; h.metallic = metallic; h.transmission = transmission; h.ior = ior;
    Vec4 r = evalBSDF(&h, from3(n), from3(v), from3(l), front);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}

int pto_raygen(const ptmi_camera *cam, uint32_t n, const uint32_t *xs, const uint32_t *ys,
               const uint32_t *frames, float *o3, float *d3, uint32_t *rng_out) {
    for (uint32_t i = 0; i < n; i++) {
        Inv inv; memset(&inv, 0, sizeof inv);
        Ray r = cameraRay(&inv, cam, xs[i], ys[i], frames[i]);
        o3[3 * i] = r.origin.x; o3[3 * i + 1] = r.origin.y; o3[3 * i + 2] = r.origin.z;
        d3[3 * i] = r.direction.x; d3[3 * i + 1] = r.direction.y; d3[3 * i + 2] = r.direction.z;
        if (rng_out) rng_out[i] = inv.rngState;
    }
    return 0;
}

static void fold(pto_stats *st, const Inv *a) {
    st->segments += a->segments; st->shadow_rays += a->shadowRays; st->nodes_visited += a->nodes;
    st->tris_tested += a->tris; st->closest_hits += a->closestHits;
    if (a->maxStack > st->max_stack) st->max_stack = a->maxStack;
}

int pto_intersect(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                  float *t, uint32_t *tri, float *u, float *v, pto_stats *st) {
    pto_stats tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel
    {
        Inv inv; memset(&inv, 0, sizeof inv); inv.scene = s;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            Ray r = { from3(o3 + 3 * i), from3(d3 + 3 * i) };
            HitInfo h = sceneIntersect(&inv, r);
            t[i] = h.t; tri[i] = h.triangle; u[i] = h.u; v[i] = h.v;
        }
#pragma omp critical
        fold(&tot, &inv);
    }
    if (st) { st->nodes_visited += tot.nodes_visited; st->tris_tested += tot.tris_tested; if (tot.max_stack > st->max_stack) st->max_stack = tot.max_stack; }
    return 0;
}

int pto_occluded(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                 const float *dist, uint8_t *occluded, pto_stats *st) {
    pto_stats tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel
    {
        Inv inv; memset(&inv, 0, sizeof inv); inv.scene = s;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            Ray r = { from3(o3 + 3 * i), from3(d3 + 3 * i) };
            HitInfo shadowHit = sceneIntersect(&inv, r);
            float d = dist ? dist[i] : -1.0f;
            if (d < 0.0f) occluded[i] = shadowHit.t > 0.0f;                                   /* :394 */
            else occluded[i] = (shadowHit.t > 0.0f && shadowHit.t < d - EPSILON * 2.0f);     /* :423, :465 */
        }
#pragma omp critical
        fold(&tot, &inv);
    }
    if (st) { st->nodes_visited += tot.nodes_visited; st->tris_tested += tot.tris_tested; }
    return 0;
}

int pto_render(const pto_scene *s, const ptmi_camera *cam, uint32_t n_frames,
               const pto_options *opt, float *out, pto_stats *st) {
    uint32_t W = cam->width, H = cam->height;
    uint32_t y0 = opt ? opt->y0 : 0u, y1 = (opt && opt->y1) ? opt->y1 : H;
    uint32_t maxBounces = opt ? opt->max_bounces : 8u;
    int doMis = opt ? (int)opt->do_mis : 1;
    if (y1 > H) y1 = H;
    if (y0 > y1) return -1;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = (opt && opt->threads) ? (int)opt->threads : omp_get_max_threads();
    double t0 = omp_get_wtime();
#endif
    pto_stats tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel num_threads(nthreads)
    {
        Inv inv; memset(&inv, 0, sizeof inv); inv.scene = s;
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = y0; y < (int64_t)y1; y++) {
            for (uint32_t x = 0; x < W; x++) {
                float *outputBuffer = out + ((size_t)y * W + x) * 4;          /* bufferIndex = y * width + x, :753 */
                for (uint32_t k = 0; k < n_frames; k++) {                     /* one dispatch per frame, renderer.ts:415-431 */
                    uint32_t frameIndex = cam->frame_index + k;
                    Ray ray = cameraRay(&inv, cam, x, (uint32_t)y, frameIndex);
                    Vec3 color = trace(&inv, ray, maxBounces, doMis);
                    color = vec3(w_min(color.x, 2.5f), w_min(color.y, 2.5f), w_min(color.z, 2.5f));
                    if (frameIndex > 0u) {
                        float t = 1.0f / (float)(frameIndex + 1u);
                        color = vec3(w_mix(outputBuffer[0], color.x, t), w_mix(outputBuffer[1], color.y, t),
                                     w_mix(outputBuffer[2], color.z, t));
                    }
                    outputBuffer[0] = color.x; outputBuffer[1] = color.y; outputBuffer[2] = color.z; outputBuffer[3] = 0.0f;
                }
            }
        }
#pragma omp critical
        fold(&tot, &inv);
    }
    if (st) {
        st->segments += tot.segments; st->shadow_rays += tot.shadow_rays; st->nodes_visited += tot.nodes_visited;
        st->tris_tested += tot.tris_tested; st->closest_hits += tot.closest_hits;
        if (tot.max_stack > st->max_stack) st->max_stack = tot.max_stack;
        st->paths += (uint64_t)(y1 - y0) * W * n_frames;
#ifdef _OPENMP
        st->seconds += omp_get_wtime() - t0;
#endif
        st->threads = (uint32_t)nthreads;
    }
    return 0;
}
