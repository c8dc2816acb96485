// traverse.hip — BVH traversal kernels of the wavefront path tracer.
//
//   extend : closest hit per queued path  (reference: src/shader/pt.wgsl:248-296,
//            Moller-Trumbore part of :123-158, slab test :234-245)
//   shadow : any-hit visibility of the next-event record written by `shade`
//            (reference: the sceneIntersect calls of pt.wgsl:392/421/463 and the
//            occlusion predicates of :394/:423/:465)
//
// Result contract: the same (t, triangle, u, v) as the reference's
// traversal — the minimum t over all triangles in leaves whose ancestors all pass
// the slab test, ties to the lowest triangle index (= first in the reference's
// left-first DFS) — reached by an ordered two-box-per-step descent with a
// conservative distance cull. cull = 0 visits exactly the reference's leaf set.
//
// Two memory variants share one traversal body:
//   global : wide nodes / triangle images read through L1/L2, per-lane stack in LDS
//   lds    : the whole traversal image staged into LDS once per persistent workgroup
#include "pt_device.h"
#include "pt_math.h"

namespace {

struct GlobalMem {
    const float4 *wn, *tp;
    PT_DEV void node(uint32_t i, float4 &a, float4 &b, float4 &c, float4 &d) const {
        const float4 *p = wn + 4u * (size_t)i;
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    PT_DEV void tri(uint32_t i, float4 &a, float4 &b, float4 &c) const {
        const float4 *p = tp + 3u * (size_t)i;
        a = p[0]; b = p[1]; c = p[2];
    }
};

PT_DEV bool slab(float bx0, float by0, float bz0, float bx1, float by1, float bz1, v3 o, v3 inv, float &tmin) {
    // pt.wgsl:234-245 with (bound - o) * (1/d)
    float t1x = (bx0 - o.x) * inv.x, t2x = (bx1 - o.x) * inv.x;
    float t1y = (by0 - o.y) * inv.y, t2y = (by1 - o.y) * inv.y;
    float t1z = (bz0 - o.z) * inv.z, t2z = (bz1 - o.z) * inv.z;
    tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= tmin && tmax >= 0.0f;
}

// pt.wgsl:128-158; returns t (> 1e-6) or -1
PT_DEV float tri_test(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float &uo, float &vo) {
    v3 h = cross3(d, e2);
    float a = dot3(e1, h);
    if (__builtin_fabsf(a) < PT_EPS) return -1.0f;
    float f = 1.0f / a;
    v3 sv = sub3(o, v0);
    float u = f * dot3(sv, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    v3 q = cross3(sv, e1);
    float v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * dot3(e2, q);
    if (t > PT_EPS) { uo = u; vo = v; return t; }
    return -1.0f;
}

// distance beyond which a box cannot hold a nearer hit; the slack covers the
// rounding difference between a slab entry distance and a triangle's own t
PT_DEV float cull_limit(float t) { return fma1(t, 1.001f, 1e-4f); }

struct Hit { float t, u, v; uint32_t tri; };

// ANYHIT: returns true at the first accepted hit with (tlim < 0 || t < tlim).
// Closest: fills `best` (t = +inf, tri = NONE when nothing is hit).
template <bool ANYHIT, bool CULL, int STACK, class Mem>
PT_DEV bool traverse(const Mem &m, const DevScene &sc, v3 o, v3 d, float tlim, uint32_t *stk, int stride, Hit &best) {
    best.t = __builtin_inff(); best.u = 0.0f; best.v = 0.0f; best.tri = PT_REF_NONE;
    uint32_t cur = sc.root_ref;
    if (cur == PT_REF_NONE) return false;
    v3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float limit = __builtin_inff();
    if (ANYHIT && CULL && !(tlim < 0.0f)) limit = cull_limit(tlim);
    float tm;
    if (!slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], o, inv, tm))
        return false;
    int sp = 0;
    for (;;) {
        if (!(cur & PT_REF_LEAF)) {
            float4 a, b, c, r;
            m.node(cur, a, b, c, r);
            float tl, tr;
            bool hl = slab(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tl);
            bool hr = slab(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tr);
            if (CULL) { hl = hl && !(tl > limit); hr = hr && !(tr > limit); }
            uint32_t lref = __float_as_uint(r.x), rref = __float_as_uint(r.y);
            if (hl && hr) {
                bool left_first = tl <= tr;
                uint32_t far = left_first ? rref : lref;
                cur = left_first ? lref : rref;
                if (sp < STACK) { stk[sp * stride] = far; sp++; }
                continue;
            }
            if (hl) { cur = lref; continue; }
            if (hr) { cur = rref; continue; }
        } else {
            uint32_t off = cur & PT_LEAF_OFF_MASK;
            uint32_t cnt = ((cur >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u;
            for (uint32_t k = 0; k < cnt; k++) {
                uint32_t ti = off + k;
                float4 a, b, c;
                m.tri(ti, a, b, c);
                float u = 0.0f, v = 0.0f;
                float t = tri_test(xyz(a), xyz(b), xyz(c), o, d, u, v);
                if (t > 0.0f) {
                    if (ANYHIT) {
                        if (tlim < 0.0f || t < tlim) return true;
                    } else if (t < best.t || (t == best.t && ti < best.tri)) {
                        // pt.wgsl:275 keeps the first strictly nearer hit of a left-first DFS:
                        // the lowest triangle index among equal t
                        best.t = t; best.u = u; best.v = v; best.tri = ti;
                        if (CULL) limit = cull_limit(t);
                    }
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stk[sp * stride];
    }
    return false;
}

PT_DEV float4 pack_hit(const Hit &h) {
    if (h.tri == PT_REF_NONE) return make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(PT_REF_NONE));
    return make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
}

// ------------------------------------------------------------------ global ----
constexpr int GBLOCK = 256;

template <int STACK, bool CULL>
__global__ __launch_bounds__(GBLOCK) void k_extend_global(DevScene sc, const float4 *__restrict__ O,
                                                          const float4 *__restrict__ D,
                                                          const uint32_t *__restrict__ queue,
                                                          const uint32_t *__restrict__ count_ptr,
                                                          float4 *__restrict__ hits) {
    __shared__ uint32_t stk[STACK * GBLOCK];
    const uint32_t count = *count_ptr;
    GlobalMem m{sc.wnodes, sc.tripos};
    for (uint32_t i = blockIdx.x * GBLOCK + threadIdx.x; i < count; i += gridDim.x * GBLOCK) {
        uint32_t p = queue ? queue[i] : i;
        float4 o4 = O[p], d4 = D[p];
        Hit h;
        traverse<false, CULL, STACK>(m, sc, xyz(o4), xyz(d4), -1.0f, stk + threadIdx.x, GBLOCK, h);
        hits[i] = pack_hit(h);
    }
}

template <int STACK, bool CULL>
__global__ __launch_bounds__(GBLOCK) void k_shadow_global(DevScene sc, DevPaths P, DevShadow S,
                                                          const uint64_t *__restrict__ mask,
                                                          const uint32_t *__restrict__ count_ptr,
                                                          uint8_t *__restrict__ occluded_out) {
    __shared__ uint32_t stk[STACK * GBLOCK];
    const uint32_t count = *count_ptr;
    GlobalMem m{sc.wnodes, sc.tripos};
    for (uint32_t i = blockIdx.x * GBLOCK + threadIdx.x; i < count; i += gridDim.x * GBLOCK) {
        if (mask && !((mask[i >> 6] >> (i & 63u)) & 1ull)) continue;
        float4 so = S.SO[i], sd = S.SD[i];
        float dist = so.w;
        float tlim = dist < 0.0f ? -1.0f : dist - PT_EPS * 2.0f;          // pt.wgsl:423, :465
        Hit h;
        bool occ = traverse<true, CULL, STACK>(m, sc, xyz(so), xyz(sd), tlim, stk + threadIdx.x, GBLOCK, h);
        if (occluded_out) { occluded_out[i] = occ ? 1 : 0; continue; }
        if (!occ) {
            uint32_t p = __float_as_uint(sd.w);
            float4 l = P.L[p], c = S.SC[i];
            P.L[p] = make_float4(l.x + c.x, l.y + c.y, l.z + c.z, 0.0f);  // pt.wgsl:675
        }
    }
}

// --------------------------------------------------------------------- LDS ----
// One persistent 1024-thread workgroup per CU stages the traversal image
// (wide nodes + triangle images) into LDS once, then walks queue chunks.
constexpr int LBLOCK = 1024;

struct LdsMem {
    const float4 *wn, *tp;      // LDS
    PT_DEV void node(uint32_t i, float4 &a, float4 &b, float4 &c, float4 &d) const {
        const float4 *p = wn + 4u * i;
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    PT_DEV void tri(uint32_t i, float4 &a, float4 &b, float4 &c) const {
        const float4 *p = tp + 3u * i;
        a = p[0]; b = p[1]; c = p[2];
    }
};

PT_DEV void stage_scene(const DevScene &sc, float4 *smem) {
    const uint32_t nw = 4u * sc.n_wnodes, nt = 3u * sc.n_tris;
    for (uint32_t i = threadIdx.x; i < nw; i += LBLOCK) smem[i] = sc.wnodes[i];
    for (uint32_t i = threadIdx.x; i < nt; i += LBLOCK) smem[nw + i] = sc.tripos[i];
    __syncthreads();
}

template <int STACK, bool CULL>
__global__ __launch_bounds__(LBLOCK) void k_extend_lds(DevScene sc, const float4 *__restrict__ O,
                                                       const float4 *__restrict__ D,
                                                       const uint32_t *__restrict__ queue,
                                                       const uint32_t *__restrict__ count_ptr,
                                                       float4 *__restrict__ hits) {
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * LBLOCK >= count) return;                 // whole workgroup has nothing to do
    stage_scene(sc, smem);
    LdsMem m{smem, smem + 4u * sc.n_wnodes};
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem + 4u * sc.n_wnodes + 3u * sc.n_tris) + threadIdx.x;
    for (uint32_t i = blockIdx.x * LBLOCK + threadIdx.x; i < count; i += gridDim.x * LBLOCK) {
        uint32_t p = queue ? queue[i] : i;
        float4 o4 = O[p], d4 = D[p];
        Hit h;
        traverse<false, CULL, STACK>(m, sc, xyz(o4), xyz(d4), -1.0f, stk, LBLOCK, h);
        hits[i] = pack_hit(h);
    }
}

template <int STACK, bool CULL>
__global__ __launch_bounds__(LBLOCK) void k_shadow_lds(DevScene sc, DevPaths P, DevShadow S,
                                                       const uint64_t *__restrict__ mask,
                                                       const uint32_t *__restrict__ count_ptr,
                                                       uint8_t *__restrict__ occluded_out) {
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * LBLOCK >= count) return;
    stage_scene(sc, smem);
    LdsMem m{smem, smem + 4u * sc.n_wnodes};
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem + 4u * sc.n_wnodes + 3u * sc.n_tris) + threadIdx.x;
    for (uint32_t i = blockIdx.x * LBLOCK + threadIdx.x; i < count; i += gridDim.x * LBLOCK) {
        if (mask && !((mask[i >> 6] >> (i & 63u)) & 1ull)) continue;
        float4 so = S.SO[i], sd = S.SD[i];
        float dist = so.w;
        float tlim = dist < 0.0f ? -1.0f : dist - PT_EPS * 2.0f;
        Hit h;
        bool occ = traverse<true, CULL, STACK>(m, sc, xyz(so), xyz(sd), tlim, stk, LBLOCK, h);
        if (occluded_out) { occluded_out[i] = occ ? 1 : 0; continue; }
        if (!occ) {
            uint32_t p = __float_as_uint(sd.w);
            float4 l = P.L[p], c = S.SC[i];
            P.L[p] = make_float4(l.x + c.x, l.y + c.y, l.z + c.z, 0.0f);
        }
    }
}

size_t lds_bytes(const TraverseConfig &cfg) {
    return cfg.lds_scene_bytes + (size_t)cfg.stack_entries * LBLOCK * sizeof(uint32_t);
}

template <int STACK, bool CULL>
void extend_dispatch(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                     const uint32_t *queue, const uint32_t *count, float4 *hits) {
    if (cfg.variant == PT_VARIANT_LDS) {
        int lb = blocks / 8; if (lb < 1) lb = 1;          // one persistent 1024-thread workgroup per CU
        hipLaunchKernelGGL((k_extend_lds<STACK, CULL>), dim3(lb), dim3(LBLOCK), lds_bytes(cfg), s, sc, p.O, p.D,
                           queue, count, hits);
    } else {
        hipLaunchKernelGGL((k_extend_global<STACK, CULL>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p.O, p.D, queue,
                           count, hits);
    }
}
template <int STACK, bool CULL>
void shadow_dispatch(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                     DevShadow sh, const uint64_t *mask, const uint32_t *count, uint8_t *occ) {
    if (cfg.variant == PT_VARIANT_LDS) {
        int lb = blocks / 8; if (lb < 1) lb = 1;          // one persistent 1024-thread workgroup per CU
        hipLaunchKernelGGL((k_shadow_lds<STACK, CULL>), dim3(lb), dim3(LBLOCK), lds_bytes(cfg), s, sc, p, sh, mask,
                           count, occ);
    } else {
        hipLaunchKernelGGL((k_shadow_global<STACK, CULL>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p, sh, mask, count,
                           occ);
    }
}

template <class F16, class F32, class F64>
void by_stack(int entries, F16 f16, F32 f32, F64 f64) {
    if (entries <= 16) f16(); else if (entries <= 32) f32(); else f64();
}

}  // namespace

int pt_extend_set_lds_limit(size_t bytes) {
    hipError_t e = hipSuccess;
#define PT_SET(K) do { hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void *>(&K), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (r != hipSuccess) e = r; } while (0)
    PT_SET((k_extend_lds<16, true>)); PT_SET((k_extend_lds<16, false>));
    PT_SET((k_extend_lds<32, true>)); PT_SET((k_extend_lds<32, false>));
    PT_SET((k_shadow_lds<16, true>)); PT_SET((k_shadow_lds<16, false>));
    PT_SET((k_shadow_lds<32, true>)); PT_SET((k_shadow_lds<32, false>));
#undef PT_SET
    return e == hipSuccess ? 0 : -1;
}

void pt_launch_extend(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      const uint32_t *queue, const uint32_t *count, float4 *hits) {
    const bool lds = cfg.variant == PT_VARIANT_LDS;
    if (cfg.cull) {
        by_stack(cfg.stack_entries,
                 [&] { extend_dispatch<16, true>(s, blocks, cfg, sc, p, queue, count, hits); },
                 [&] { extend_dispatch<32, true>(s, blocks, cfg, sc, p, queue, count, hits); },
                 [&] { if (lds) extend_dispatch<32, true>(s, blocks, cfg, sc, p, queue, count, hits);
                       else hipLaunchKernelGGL((k_extend_global<64, true>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p.O,
                                               p.D, queue, count, hits); });
    } else {
        by_stack(cfg.stack_entries,
                 [&] { extend_dispatch<16, false>(s, blocks, cfg, sc, p, queue, count, hits); },
                 [&] { extend_dispatch<32, false>(s, blocks, cfg, sc, p, queue, count, hits); },
                 [&] { if (lds) extend_dispatch<32, false>(s, blocks, cfg, sc, p, queue, count, hits);
                       else hipLaunchKernelGGL((k_extend_global<64, false>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p.O,
                                               p.D, queue, count, hits); });
    }
}

void pt_launch_shadow(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      DevShadow sh, const uint64_t *mask, const uint32_t *count, uint8_t *occ) {
    const bool lds = cfg.variant == PT_VARIANT_LDS;
    if (cfg.cull) {
        by_stack(cfg.stack_entries,
                 [&] { shadow_dispatch<16, true>(s, blocks, cfg, sc, p, sh, mask, count, occ); },
                 [&] { shadow_dispatch<32, true>(s, blocks, cfg, sc, p, sh, mask, count, occ); },
                 [&] { if (lds) shadow_dispatch<32, true>(s, blocks, cfg, sc, p, sh, mask, count, occ);
                       else hipLaunchKernelGGL((k_shadow_global<64, true>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p, sh,
                                               mask, count, occ); });
    } else {
        by_stack(cfg.stack_entries,
                 [&] { shadow_dispatch<16, false>(s, blocks, cfg, sc, p, sh, mask, count, occ); },
                 [&] { shadow_dispatch<32, false>(s, blocks, cfg, sc, p, sh, mask, count, occ); },
                 [&] { if (lds) shadow_dispatch<32, false>(s, blocks, cfg, sc, p, sh, mask, count, occ);
                       else hipLaunchKernelGGL((k_shadow_global<64, false>), dim3(blocks), dim3(GBLOCK), 0, s, sc, p, sh,
                                               mask, count, occ); });
    }
}
