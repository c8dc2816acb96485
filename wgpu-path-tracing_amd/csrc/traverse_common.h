// traverse_common.h — what the two traversal translation units share: traverse.hip (the reference's leaves, ptmi_options.leaves = 1)
// and traverse_own.hip (the library's own leaves, leaves = 2): scheduling constants, address-space-qualified access, the contract's
// slab test (pt.wgsl:234-245), the distance cull, and the ray sources / result sinks of the closest-hit and any-hit kernels.
#pragma once
#include "pt_device.h"
#include "pt_math.h"

namespace {

constexpr int MODE_EXTEND = 0, MODE_SHADOW = 1;
#ifndef PT_REFILL_AT
#define PT_REFILL_AT 36
#endif
constexpr int REFILL_AT = PT_REFILL_AT;   // refill when at most this many of the 64 lanes still hold a ray (scene in LDS)
// The kernels that walk the scene from global memory refill earlier: a lane without a ray also means a memory request
// less in flight. Measured on the 1 M-triangle scene (Msamples/s): 28: 4 404, 36: 4 543, 44: 4 636, 52: 4 667, 58: 4 651; Cornell ±1 % throughout.
#ifndef PT_REFILL_GLOBAL
#define PT_REFILL_GLOBAL 52
#endif
// One vote (two ballots, the refill and completion tests) costs about half a box-pair step, so a stream keeps
// running for up to NODE_STEPS steps / LEAF_STEPS leaves while enough of the lanes that started it can go on:
// it stops when fewer than 1/NODE_KEEP (1/LEAF_KEEP) of them remain. Measured per kernel on Cornell 1080p.
#ifndef PT_NODE_STEPS
#define PT_NODE_STEPS 8
#endif
#ifndef PT_LEAF_STEPS
#define PT_LEAF_STEPS 4
#endif
#ifndef PT_LEAF_KEEP
#define PT_LEAF_KEEP 3
#endif
constexpr int NODE_STEPS = PT_NODE_STEPS, LEAF_STEPS = PT_LEAF_STEPS, LEAF_KEEP = PT_LEAF_KEEP;
// The box-step loop is unrolled NODE_STEPS times in the kernels whose stacks live entirely in LDS. The spilling variants (scenes walked
// from global memory, mid-size trees) carry the spill and un-spill paths in every copy, twice (the streams exist in two copies): their
// unroll count is a parameter of its own (measured: profiles/README.md, round 3)
#ifndef PT_SPILL_NODE_UNROLL
#define PT_SPILL_NODE_UNROLL 8
#endif

// Loads go through address-space-qualified pointers so that the compiler emits ds_read_b128 /
// global_load_dwordx4 and never a FLAT load: with generic pointers it merged the LDS read of a node
// with the (rare) global read of the uploaded tree into one flat_load of a selected address.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f4v *lds_f4p;
typedef const __attribute__((address_space(1))) f4v *glb_f4p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
PT_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the EXEC-masked lane mask of a predicate, straight from the compare (HIP's __ballot goes through a VGPR 0/1 value)
PT_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// 32-bit forms of two tests the compiler otherwise does in 64 bits (on the VALU): a mask's popcount as an int,
// and "at least two free entries between the node stack and the leaf list" (a negative difference means none)
PT_DEV int popc(uint64_t m) { return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32)); }
PT_DEV bool room2(lds_u32p lp, lds_u32p sp, int stride) {
    return (int)((uint32_t)(uintptr_t)lp - (uint32_t)(uintptr_t)sp) >= stride * 4;
}
PT_DEV float4 as_f4(f4v v) { return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV void load_node(glb_f4p p, float4 &a, float4 &b, float4 &c, float4 &d) {
    a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); d = as_f4(p[3]);
}

// ---- node / leaf access policies ---------------------------------------------------------------------------------
// node(i, old, b)   : both child boxes and child references of wide node i
// open(ref, ...)    : a filed leaf -> its triangle range (first index, count) and a cursor for tri(); false = skip it
// tri(cursor, k, ..): v0, e1, e2 of the leaf's k-th triangle
// `old` marks a lane that walks the tree exactly as uploaded (irregular rays): only the quantised
// variant stores that tree in a different format than the one it normally walks.
struct Boxes { float lx0, ly0, lz0, lx1, ly1, lz1, rx0, ry0, rz0, rx1, ry1, rz1; uint32_t lref, rref; };
PT_DEV void boxes_of(float4 a, float4 b, float4 c, float4 r, Boxes &o) {
    o.lx0 = a.x; o.ly0 = a.y; o.lz0 = a.z; o.lx1 = a.w; o.ly1 = b.x; o.lz1 = b.y;
    o.rx0 = b.z; o.ry0 = b.w; o.rz0 = c.x; o.rx1 = c.y; o.ry1 = c.z; o.rz1 = c.w;
    o.lref = __float_as_uint(r.x); o.rref = __float_as_uint(r.y);
}
PT_DEV void open_plain(uint32_t ref, uint32_t &first, uint32_t &cnt, uint32_t &cursor) {
    first = ref & PT_LEAF_OFF_MASK; cnt = ((ref >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u; cursor = first;
}

// -DPT_UTIL_STATS (a diagnostic build, tools/lane_stats.py; never the shipped library): where a wave's lanes idle. Per kernel
// kind: [0] votes, [1] lanes holding a ray at the vote, [2] refills, [3] lanes refilled, [4] box-pair steps, [5] lanes taking
// part, [6] leaves opened (wave steps), [7] lanes opening one, [8] triangle iterations, [9] lanes testing a triangle
#ifdef PT_UTIL_STATS
static __device__ unsigned long long g_util[2][16];     // one per translation unit (no relocatable device code)
#define UTIL(i, v) (ut[i] += (uint32_t)(v))
#else
#define UTIL(i, v) ((void)0)
#endif

PT_DEV bool slab(float bx0, float by0, float bz0, float bx1, float by1, float bz1, v3 o, v3 inv, float &tmin) {
    // pt.wgsl:234-245 with (bound - o) * (1/d)
    float t1x = (bx0 - o.x) * inv.x, t2x = (bx1 - o.x) * inv.x;
    float t1y = (by0 - o.y) * inv.y, t2y = (by1 - o.y) * inv.y;
    float t1z = (bz0 - o.z) * inv.z, t2z = (bz1 - o.z) * inv.z;
    tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= tmin && tmax >= 0.0f;
}

// distance beyond which a box cannot hold a nearer hit; the slack covers the
// rounding difference between a slab entry distance and a triangle's own t
PT_DEV float cull_limit(float t) { return fma1(t, 1.001f, 1e-4f); }

struct Hit { float t; uint32_t tri; };      // (u, v) are not kept: `shade` rebuilds them from the triangle (pt_math.h tri_test)

PT_DEV float2 pack_hit(const Hit &h) {
    if (h.tri == PT_REF_NONE) return make_float2(-1.0f, __uint_as_float(PT_REF_NONE));
    return make_float2(h.t, __uint_as_float(h.tri));
}

// ---- ray sources / result sinks of the two kernels ---------------------------------
struct ExtendIO {
    const float4 *O, *D; const uint32_t *queue; float2 *hits;
    PT_DEV bool fetch(uint32_t slot, v3 &o, v3 &d, float &tlim) const {
        uint32_t p = queue ? queue[slot] : slot;
        float4 o4 = O[p], d4 = D[p];
        o = xyz(o4); d = xyz(d4); tlim = 0.0f;
        return true;
    }
    PT_DEV void finish(uint32_t slot, const Hit &h, bool) const { st_stream(&hits[slot], pack_hit(h)); }
    // (traverse_own.hip: `aux` is a word a lane keeps from fetch to finish — the any-hit kernel's path id; nothing here)
    PT_DEV bool fetch(uint32_t slot, v3 &o, v3 &d, float &tlim, uint32_t &) const { return fetch(slot, o, d, tlim); }
    PT_DEV void finish(uint32_t slot, const Hit &h, bool occ, uint32_t) const { finish(slot, h, occ); }
};
// The records of a bounce are one allocation (pt_device.h DevShadow): the kernel keeps its base and `cap` instead of three
// pointers, and the radiance buffer instead of the whole path state — the node-cache variant needs at most 80 scalar
// registers for its two workgroups per CU.
struct ShadowIO {
    float *L; const float4 *rec; const uint32_t *sq; uint32_t l_stride, cap;
    // false: nothing to trace — the record of an emissive hit (SO.w = -2, shade.hip), added to L like an unoccluded sample
    PT_DEV bool fetch(uint32_t &slot, v3 &o, v3 &d, float &tlim) const {
        uint32_t i = sq ? sq[slot] : slot;
        slot = i;                                              // the record's own slot is what finish() needs
        float4 so = ld_stream(&rec[i]), sd = ld_stream(&rec[(size_t)cap + i]);
        o = xyz(so); d = xyz(sd);
        // pt.wgsl:423, :465: occluded iff a hit is nearer than dist - 2e-6 (negative for a light closer than 2e-6: never
        // occluded). A directional light (:394) has no distance, any hit occludes, one at t = +inf included: tlim = NaN,
        // and the tests below are written so that NaN means "no limit" (!(t >= NaN) is true, tl > NaN is false).
        tlim = so.w < 0.0f ? __builtin_nanf("") : so.w - PT_EPS * 2.0f;
        return so.w != -2.0f;
    }
    PT_DEV void finish(uint32_t i, const Hit &h, bool occluded) const {
        if (!occluded) finish(i, h, occluded, __float_as_uint(rec[(size_t)cap + i].w));
    }
    // the path id travels in the lane from fetch (it is SD.w of the record fetch reads anyway) to finish: the radiance and the
    // contribution are then loaded together instead of one after a reload of the record
    PT_DEV bool fetch(uint32_t &slot, v3 &o, v3 &d, float &tlim, uint32_t &path) const {
        uint32_t i = sq ? sq[slot] : slot;
        slot = i;
        float4 so = ld_stream(&rec[i]), sd = ld_stream(&rec[(size_t)cap + i]);
        o = xyz(so); d = xyz(sd); path = __float_as_uint(sd.w);
        tlim = so.w < 0.0f ? __builtin_nanf("") : so.w - PT_EPS * 2.0f;
        return so.w != -2.0f;
    }
    PT_DEV void finish(uint32_t i, const Hit &, bool occluded, uint32_t p) const {
        if (!occluded) {
            DevPaths P; P.O = nullptr; P.D = nullptr; P.C = nullptr; P.L = L; P.l_stride = l_stride;
            const rgb_sc l = P.ldL(p), c = reinterpret_cast<const rgb_sc *>(rec + 2 * (size_t)cap)[i];
            P.stL(p, l.x + c.x, l.y + c.y, l.z + c.z);   // pt.wgsl:675
        }
    }
};
// ptmi_debug_occluded: the same rays, the verdict written out instead of added
struct OccludedIO {
    const float4 *rec; uint8_t *occluded_out; uint32_t cap;
    PT_DEV bool fetch(uint32_t &slot, v3 &o, v3 &d, float &tlim) const {
        ShadowIO s{nullptr, rec, nullptr, 3u, cap};
        return s.fetch(slot, o, d, tlim);
    }
    PT_DEV void finish(uint32_t i, const Hit &, bool occluded) const { occluded_out[i] = occluded ? 1 : 0; }
    PT_DEV bool fetch(uint32_t &slot, v3 &o, v3 &d, float &tlim, uint32_t &) const { return fetch(slot, o, d, tlim); }
    PT_DEV void finish(uint32_t i, const Hit &h, bool occluded, uint32_t) const { finish(i, h, occluded); }
};

}  // namespace
