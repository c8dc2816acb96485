// traverse.hip — BVH traversal kernels of the wavefront path tracer.
//
//   extend : closest hit per queued path  (reference: src/shader/pt.wgsl:248-296,
//            Moller-Trumbore part of :123-158, slab test :234-245)
//   shadow : any-hit visibility of the next-event record written by `shade`
//            (reference: the sceneIntersect calls of pt.wgsl:392/421/463 and the
//            occlusion predicates of :394/:423/:465)
//
// Result contract: the same (t, triangle) — and with them the same (u, v), which `shade` recomputes —
// as the reference's traversal — the minimum t over all triangles in leaves whose ancestors all pass
// the slab test, ties to the lowest triangle index (= first in the reference's
// left-first DFS) — reached by an ordered two-box-per-step descent with a
// conservative distance cull, over a hierarchy rebuilt on the reference's leaves
// (fast_tree.hip) or the tree exactly as uploaded. cull = 0 tests exactly the
// reference's leaf set.
//
// Execution model: PERSISTENT WAVES. A wave owns a strided share of the ray queue and
// keeps 64 rays in flight; when enough lanes have finished their ray, the idle lanes
// fetch the wave's next rays (rank by ballot/popcount, no atomics), so a wave is not
// held hostage by its slowest ray. Per-ray work on Cornell varies widely (measured:
// max-over-64 / mean = 2.0), which is what this removes.
//
// Memory variants that share the body:
//   global     : wide nodes / triangle images read through L1/L2; 16 stack entries per lane in LDS, deeper node
//                stacks spill to global memory (any depth up to the 62 the upload accepts). Scenes beyond an L2 walk
//                the QUANTISED image (32-byte nodes, leaf stream with the exact leaf boxes, top of the tree in LDS: QuantMem)
//   lds        : the whole traversal image staged into LDS once per persistent workgroup
//   node cache : only the wide nodes staged — small trees: two workgroups share a CU; mid-size trees: one
//                workgroup per CU with spilling stacks
#include "pt_device.h"
#include "pt_math.h"
#include <atomic>
#include <type_traits>

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

PT_DEV bool slab(float bx0, float by0, float bz0, float bx1, float by1, float bz1, v3 o, v3 inv, float &tmin);

// -DPT_UTIL_STATS (a diagnostic build, tools/lane_stats.py; never the shipped library): where a wave's lanes idle. Per kernel
// kind: [0] votes, [1] lanes holding a ray at the vote, [2] refills, [3] lanes refilled, [4] box-pair steps, [5] lanes taking
// part, [6] leaves opened (wave steps), [7] lanes opening one, [8] triangle iterations, [9] lanes testing a triangle
#ifdef PT_UTIL_STATS
__device__ unsigned long long g_util[2][16];
#define UTIL(i, v) (ut[i] += (uint32_t)(v))
#else
#define UTIL(i, v) ((void)0)
#endif

struct GlobalMem {
    glb_f4p wn, tp;
    PT_DEV void node(uint32_t i, bool, Boxes &o) const { float4 a, b, c, r; load_node(wn + 4u * (size_t)i, a, b, c, r); boxes_of(a, b, c, r, o); }
    PT_DEV bool open(uint32_t ref, bool, v3, v3, float, uint32_t &first, uint32_t &cnt, uint32_t &cursor) const {
        open_plain(ref, first, cnt, cursor); return true;
    }
    PT_DEV void tri(uint32_t cursor, uint32_t k, bool, float4 &a, float4 &b, float4 &c) const {
        glb_f4p p = tp + 3u * (size_t)(cursor + k);
        a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]);
    }
};
// wide nodes in LDS; triangle images in LDS too (TRIS_IN_LDS) or read through L1/L2 (the node cache:
// half the LDS, so two workgroups fit a CU)
template <bool TRIS_IN_LDS>
struct LdsMem {
    lds_f4p wn, tl; glb_f4p tg;
    PT_DEV void node(uint32_t i, bool, Boxes &o) const {
        lds_f4p p = wn + 4u * i;
        boxes_of(as_f4(p[0]), as_f4(p[1]), as_f4(p[2]), as_f4(p[3]), o);
    }
    PT_DEV bool open(uint32_t ref, bool, v3, v3, float, uint32_t &first, uint32_t &cnt, uint32_t &cursor) const {
        open_plain(ref, first, cnt, cursor); return true;
    }
    PT_DEV void tri(uint32_t cursor, uint32_t k, bool, float4 &a, float4 &b, float4 &c) const {
        const uint32_t i = cursor + k;
        if (TRIS_IN_LDS) { lds_f4p p = tl + 3u * i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
        else { glb_f4p p = tg + 3u * (size_t)i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
    }
};

// QUANTISED variant (large scenes, walked from global memory; fast_tree.hip::pt_quantize_tree builds the image):
//   node, 32 B = 2 x uint4: per child three words of 16-bit plane numbers (lo.x | lo.y << 16, lo.z | hi.x << 16,
//         hi.y | hi.z << 16) and its reference; plane k on axis a is fma(scale[a], k, origin[a]) — the child's exact box
//         rounded OUTWARD to that grid. A box that contains a leaf's box passes whenever the leaf's own box passes (the
//         slab predicate is monotone under containment), so the descent never loses a leaf the reference
//         would test; it may reach a few it would not, and those stop at
//   leaf stream (dwords): per leaf a header (exact lo.xyz, first triangle, exact hi.xyz, count) — the box the reference
//         tests for this leaf, tested here when the leaf is opened, so the triangles tested are exactly the reference's —
//         followed by 9 dwords per triangle (v0, e1, e2). A leaf reference is PT_REF_LEAF | dword offset of its header.
// Half the bytes per node step and 25 % fewer per triangle: this variant waits on L2 / Infinity Cache as much as on the ALUs.
// Lanes that walk the tree as uploaded (`old`) read the exact 64-B image and the 48-B triangle images instead.
//
// The decoded planes go through the contract's own slab arithmetic. (Tried: one fused step per plane,
// fma(k, scale * inv, (origin - o) * inv), with each axis' interval widened by a proven bound of its rounding error — any
// test that passes whenever the contract's passes would do for inner boxes. 24 VALU instructions fewer per node, 9 more
// registers per ray: 5 waves per SIMD instead of 6, extend -3 %, shadow +11 % on the 1 M-triangle scene. The kernel waits on
// node fetches as much as on the ALUs; registers are worth more than instructions here.)
typedef const __attribute__((address_space(1))) uint32_t *glb_u32p;
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u4v *glb_u4p;
typedef const __attribute__((address_space(3))) u4v *lds_u4p;
struct QuantMem {
    glb_u4p qn; glb_u32p ls; glb_f4p tp;
    float ox, oy, oz, sx, sy, sz;
    lds_u4p qc; uint32_t n_cached;        // the first n_cached nodes (the top levels, numbered breadth-first) also live in LDS
    PT_DEV void node(uint32_t i, bool, Boxes &b) const {
        u4v l, r;
        if (i < n_cached) { l = qc[2u * i]; r = qc[2u * i + 1u]; }
        else { l = qn[2u * (size_t)i]; r = qn[2u * (size_t)i + 1u]; }
        b.lx0 = fma1(sx, (float)(l.x & 0xFFFFu), ox); b.ly0 = fma1(sy, (float)(l.x >> 16), oy);
        b.lz0 = fma1(sz, (float)(l.y & 0xFFFFu), oz); b.lx1 = fma1(sx, (float)(l.y >> 16), ox);
        b.ly1 = fma1(sy, (float)(l.z & 0xFFFFu), oy); b.lz1 = fma1(sz, (float)(l.z >> 16), oz);
        b.rx0 = fma1(sx, (float)(r.x & 0xFFFFu), ox); b.ry0 = fma1(sy, (float)(r.x >> 16), oy);
        b.rz0 = fma1(sz, (float)(r.y & 0xFFFFu), oz); b.rx1 = fma1(sx, (float)(r.y >> 16), ox);
        b.ry1 = fma1(sy, (float)(r.z & 0xFFFFu), oy); b.rz1 = fma1(sz, (float)(r.z >> 16), oz);
        b.lref = l.w; b.rref = r.w;
    }
    PT_DEV bool open(uint32_t ref, bool plain, v3 o, v3 inv, float limit, uint32_t &first, uint32_t &cnt, uint32_t &cursor) const {
        if (plain) { open_plain(ref, first, cnt, cursor); return true; }
        const uint32_t off = ref & ~PT_REF_LEAF;
        glb_u32p h = ls + off;
        const float lx = __uint_as_float(h[0]), ly = __uint_as_float(h[1]), lz = __uint_as_float(h[2]);
        first = h[3];
        const float hx = __uint_as_float(h[4]), hy = __uint_as_float(h[5]), hz = __uint_as_float(h[6]);
        cnt = h[7];
        cursor = off + 8u;
        float tl;
        const bool pass = slab(lx, ly, lz, hx, hy, hz, o, inv, tl);       // the reference's own test of this leaf (pt.wgsl:266)
        return pass & !(tl > limit);                                       // and the distance cull, against today's limit
    }
    PT_DEV void tri(uint32_t cursor, uint32_t k, bool plain, float4 &a, float4 &b, float4 &c) const {
        if (plain) {
            glb_f4p p = tp + 3u * (size_t)(cursor + k);
            a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]);
        } else {
            glb_u32p p = ls + cursor + 9u * k;
            a = make_float4(__uint_as_float(p[0]), __uint_as_float(p[1]), __uint_as_float(p[2]), 0.0f);
            b = make_float4(__uint_as_float(p[3]), __uint_as_float(p[4]), __uint_as_float(p[5]), 0.0f);
            c = make_float4(__uint_as_float(p[6]), __uint_as_float(p[7]), __uint_as_float(p[8]), 0.0f);
        }
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
    PT_DEV void finish(uint32_t i, const Hit &, bool occluded) const {
        if (!occluded) {
            const uint32_t p = __float_as_uint(rec[(size_t)cap + i].w);
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
};

// One wave traces the 64-slot groups gw, gw + total_waves, gw + 2 total_waves, ... of a queue of
// `count` slots (the same interleaving a grid-stride loop gives, so every wave sees a uniform
// sample of the queue and the waves finish together). Virtual index v of the wave maps to slot
// ((v >> 6) * total_waves + gw) * 64 + (v & 63). stk: this lane's LDS entries, `stride` apart:
// the node stack grows from entry 0, the list of filed leaves from entry STACK-1.
//
// MAJORITY SCHEDULING: per iteration the wave runs ONE of two instruction streams — the two-box step
// or a leaf's triangle tests — whichever more of its lanes can take part in; the others keep their
// state (an if/if loop pays both streams every iteration while each lane uses one).
// DEFERRED LEAVES: a lane that finds a leaf does not wait for the wave to switch to the triangle
// stream; it files the leaf and keeps descending. Most lanes then have both kinds of work pending and
// can join whichever stream runs (lane utilisation of VALU instructions 31 % -> 65 %). The tested leaf
// set is unchanged; only the visiting order differs, and the result is order-independent
// (min over (t, index)).
// SPILL (global variant): the LDS entries hold only the top of the node stack. When a push finds no room the lane
// moves its whole LDS node stack to its column of `spill` (entry k of lane g at spill[k * spill_lanes + g], so a wave's
// accesses coalesce) and goes on with an empty one; when the LDS part runs dry it takes the last 8 spilled entries
// back. Deep trees then need no deeper LDS stacks — the occupancy of a depth-60 scene is that of a depth-14 one — and
// the order in which nodes are visited, hence every result, is unchanged.
// -DPT_DYNAMIC_CLAIM=1 (an experiment, profiles/README.md round 3): instead of the fixed share above a wave CLAIMS chunks of consecutive
// slots with one atomic each (about 8 chunks per wave of a full grid), so that a workgroup that becomes resident late — because another
// kernel holds part of the machine when this one launches — finds less work left instead of a full share to run as a second round.
#ifndef PT_DYNAMIC_CLAIM
#define PT_DYNAMIC_CLAIM 0
#endif
template <int MODE, bool CULL, int STACK, bool SPILL, int REFILL, class Mem, class IO>
PT_DEV void trace_wave(const Mem &m, const DevScene &sc, const IO &io, uint32_t count, uint32_t gw,
                             uint32_t total_waves, uint32_t *stk, int stride, uint32_t *spill = nullptr,
                             uint32_t spill_lanes = 0, uint32_t *ticket = nullptr) {
    constexpr bool ANY = MODE == MODE_SHADOW;
    constexpr int NODE_KEEP = ANY ? 2 : 3;
    const uint32_t lane = threadIdx.x & 63u;
    // gw (and so end, next) is the same in all 64 lanes; readfirstlane tells the compiler, which then keeps
    // the queue bookkeeping in SGPRs and turns the refill / exit tests into scalar branches
    gw = uniform(gw);
#if PT_DYNAMIC_CLAIM
    // the slots [cpos, cend) of the claimed chunk are still to be handed out; cpos = PT_REF_NONE: the ticket counter has run out.
    // (Two scalar registers and the counter's address: the kernels that share a CU between two workgroups have 80 in all.)
    uint32_t cpos = 0u, cend = 0u;
#else
    const uint32_t ngroups = (count + 63u) >> 6;
    const uint32_t end = gw < ngroups ? ((ngroups - gw + total_waves - 1u) / total_waves) * 64u : 0u;
    uint32_t next = 0u;
#endif
    bool active = false, slow = false;      // slow: an irregular ray or an unbounded determinant, see the refill
    const bool has_fast = sc.has_fast != 0u;
    uint32_t slot = 0, cur = PT_REF_NONE;
    // this lane's LDS entries as two pointers: the node stack grows up from `bot` (sp = next free entry), the list
    // of filed leaves grows down from `top` (lp = next free entry); STACK - used = (lp - sp) / stride + 1 entries free
    const lds_u32p bot = (lds_u32p)stk, top = bot + (STACK - 1) * stride;
    lds_u32p sp = bot, lp = top;
    uint32_t spn = 0;                       // SPILL: entries of this lane in the spill area
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 0);
    float tlim = 0.0f, limit = __builtin_inff();
    Hit best; best.t = __builtin_inff(); best.tri = PT_REF_NONE;
#ifdef PT_UTIL_STATS
    uint32_t ut[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (;;) {
        uint64_t act = ballot(active);
#if PT_DYNAMIC_CLAIM
        if (cpos >= cend && cpos != PT_REF_NONE && popc(act) <= REFILL) {      // claim the next chunk: one atomic for the wave
            uint32_t chunk = uniform(((count / (total_waves * 8u)) + 63u) & ~63u);      // about 8 chunks per wave of a full grid, at least 256 slots
            chunk = chunk < 256u ? 256u : chunk;
            uint32_t t = 0u;
            if (lane == 0u) t = atomicAdd(ticket, 1u);
            t = uniform(t);
            const unsigned long long base = (unsigned long long)t * chunk;
            if (base >= count) { cpos = PT_REF_NONE; cend = 0u; }
            else { cpos = (uint32_t)base; cend = (base + chunk < count) ? (uint32_t)(base + chunk) : count; }
        }
        if (cpos < cend && popc(act) <= REFILL) {
            const uint64_t idle = ~act;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t vslot = cpos + rank;
            if (!active && vslot < cend) {
#else
        if (next < end && popc(act) <= REFILL) {
            const uint64_t idle = ~act;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t vi = next + rank;
            const uint32_t vslot = ((vi >> 6) * total_waves + gw) * 64u + (vi & 63u);
            if (!active && vi < end && vslot < count) {
#endif
                slot = vslot;
                const bool want = io.fetch(slot, o, d, tlim);
                inv = mk3(rcp1(d.x), rcp1(d.y), rcp1(d.z));
                best.t = __builtin_inff(); best.tri = PT_REF_NONE;
                sp = bot; lp = top; spn = 0u; cur = PT_REF_NONE;
                limit = (ANY && CULL) ? cull_limit(tlim) : __builtin_inff();      // NaN for a directional light: never culls
                const bool regular = __builtin_isfinite(inv.x) & __builtin_isfinite(inv.y) & __builtin_isfinite(inv.z) &
                                     (inv.x != 0.0f) & (inv.y != 0.0f) & (inv.z != 0.0f);
                // bounded: the triangle test's determinant stays below 2^100 for this ray (NaN compares false): pt_math.h tri_test_t.
                // ONE flag for both kinds of special ray (a second wave-wide mask would take the kernel's scalar registers past 80
                // and with them the second workgroup per CU): they run the careful copy of the streams below, and where the
                // hierarchy was rebuilt they walk the tree as uploaded.
                const bool bounded = (__builtin_fabsf(d.x) + __builtin_fabsf(d.y) + __builtin_fabsf(d.z)) <= sc.tri_safe_dsum;
                slow = !(regular & bounded);
                float tm;
                if (want && sc.root_ref != PT_REF_NONE &&
                    slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2],
                         o, inv, tm)) {
                    active = true;
                    const uint32_t r = (has_fast & slow) ? sc.ref_root_ref : sc.root_ref;
                    if (r & PT_REF_LEAF) { *lp = r; lp -= stride; }              // a one-leaf tree: file the root
                    else cur = r;
                } else {
                    io.finish(slot, best, false);
                }
            }
#if PT_DYNAMIC_CLAIM
            cpos += (uint32_t)__popcll(idle);
            cpos = cpos < cend ? cpos : cend;
#else
            next += (uint32_t)__popcll(idle);
#endif
            UTIL(2, 1); UTIL(3, popc(ballot(active)) - popc(act));
            act = ballot(active);
        }
#if PT_DYNAMIC_CLAIM
        if (act == 0ull) { if (cpos == PT_REF_NONE) break; if (cpos >= cend) continue; }
#else
        if (act == 0ull && next >= end) break;
#endif
        UTIL(0, 1); UTIL(1, popc(act));

        // two entries free (a step files at most two entries) — or, with SPILL, two free once the node entries are moved out
        const bool can_node = active & (cur != PT_REF_NONE) & ((int)room2(lp, sp, stride) | (int)(SPILL && (sp != bot) & room2(lp, bot, stride)));
        const bool can_tri = active & (lp != top);
        const uint64_t bn = ballot(can_node), bt = ballot(can_tri);
        const bool run_tri = popc(bt) > popc(bn);
        bool occluded = false;
        // Two copies of the streams: lanes that walk the uploaded tree (irregular rays, use_ref) read it from global memory
        // in its own format; a wave holds such a lane almost never, and every other time it runs the copy without that
        // per-step choice. The same copy serves rays whose triangle-test determinant is not known to stay below 2^100 (not `bounded`:
        // direction components summing to more than DevScene::tri_safe_dsum — none in a dispatch, whose directions are unit
        // vectors, unless the scene has edges longer than 2^49): it keeps the range test of the short reciprocal, the other drops it.
        auto streams = [&](auto with_ref) {
        constexpr bool REF = decltype(with_ref)::value;
        const bool old = REF && has_fast && slow;
        if (run_tri) {
            bool ct = can_tri;
#pragma unroll 1
            for (int rep = 0; rep < LEAF_STEPS; rep++) {
                UTIL(6, 1); UTIL(7, popc(ballot(ct)));
                if (ct) {
                    lp += stride;                                       // next filed leaf
                    uint32_t first, cnt, cursor;
                    const bool plain = old;          // (a one-leaf tree is never quantised: its root reference is a plain one too)
                    if (!m.open(*lp, plain, o, inv, (CULL && !ANY) ? limit : __builtin_inff(), first, cnt, cursor)) cnt = 0u;
                    for (uint32_t k = 0; k < cnt; k++) {                // pt.wgsl:272-279
                        UTIL(8, uniform(lane) == lane ? 1 : 0); UTIL(9, 1);   // per-lane counts, summed at the end
                        float4 a, b, c;
                        m.tri(cursor, k, plain, a, b, c);
                        float u = 0.0f, v = 0.0f;
                        const float t = tri_test_t<!REF>(xyz(a), xyz(b), xyz(c), o, d, u, v);
                        const bool hit = t > 0.0f;
                        const uint32_t ti = first + k;
                        if (ANY) {
                            occluded = occluded | (hit & !(t >= tlim));
                        } else {
                            const bool better = hit & ((t < best.t) | ((t == best.t) & (ti < best.tri)));
                            best.t = better ? t : best.t; best.tri = better ? ti : best.tri;
                            if (CULL) limit = better ? cull_limit(t) : limit;
                        }
                    }
                }
                if (rep + 1 < LEAF_STEPS) {
                    ct = ct & (lp != top) & !occluded;
                    if (popc(ballot(ct)) * LEAF_KEEP < popc(bt)) break;
                }
            }
        } else {
            // NODE_STEPS box-pair steps per vote: the vote and the bookkeeping around it cost about half a step
            bool cn = can_node;
            constexpr int NODE_UNROLL = SPILL ? PT_SPILL_NODE_UNROLL : NODE_STEPS;
#pragma unroll NODE_UNROLL
            for (int rep = 0; rep < NODE_STEPS; rep++) {
                UTIL(4, 1); UTIL(5, popc(ballot(cn)));
                if (cn) {
                    if (SPILL && !room2(lp, sp, stride)) {              // rare: move the LDS node stack out
                        for (lds_u32p q = bot; q != sp; q += stride) { spill[(size_t)spn * spill_lanes] = *q; spn++; }
                        sp = bot;
                    }
                    float tl, tr;
                    Boxes nb;
                    if (REF && has_fast && slow) {
                        float4 a, b, c, r;
                        load_node((glb_f4p)sc.ref_wnodes + 4u * (size_t)cur, a, b, c, r);
                        boxes_of(a, b, c, r, nb);
                    } else {
                        m.node(cur, false, nb);
                    }
                    bool hl = slab(nb.lx0, nb.ly0, nb.lz0, nb.lx1, nb.ly1, nb.lz1, o, inv, tl);
                    bool hr = slab(nb.rx0, nb.ry0, nb.rz0, nb.rx1, nb.ry1, nb.rz1, o, inv, tr);
                    const uint32_t lref = nb.lref, rref = nb.rref;
                    if (CULL) { hl = hl & !(tl > limit); hr = hr & !(tr > limit); }
                    const bool ll = (lref & PT_REF_LEAF) != 0u, rl = (rref & PT_REF_LEAF) != 0u;
                    if (hl & ll) { *lp = lref; lp -= stride; }
                    if (hr & rl) { *lp = rref; lp -= stride; }
                    const bool il = hl & !ll, ir = hr & !rl;
                    const bool left_first = tl <= tr;
                    if (il & ir) { *sp = left_first ? rref : lref; sp += stride; cur = left_first ? lref : rref; }
                    else if (il) cur = lref;
                    else if (ir) cur = rref;
                    else if (sp != bot) { sp -= stride; cur = *sp; }
                    else if (SPILL && spn != 0u) {                      // rare: take the last 8 spilled entries back
                        // as many as fit under the leaf list while leaving two entries free (at least the one that is popped)
                        // (lp can sit one entry below bot when filed leaves fill the LDS entries: free = 0)
                        const int fit = (int)((uint32_t)(uintptr_t)lp - (uint32_t)(uintptr_t)bot) / (stride * 4);   // free - 1
                        uint32_t n = spn < 8u ? spn : 8u;
                        n = (int)n < fit ? n : (fit > 1 ? (uint32_t)fit : 1u);
                        spn -= n;
                        for (uint32_t j = 0; j + 1u < n; j++) { *sp = spill[(size_t)(spn + j) * spill_lanes]; sp += stride; }
                        cur = spill[(size_t)(spn + n - 1u) * spill_lanes];
                    }
                    else cur = PT_REF_NONE;
                }
                if (rep + 1 < NODE_STEPS) {
                    cn = cn & (cur != PT_REF_NONE) & ((int)room2(lp, sp, stride) | (int)(SPILL && (sp != bot) & room2(lp, bot, stride)));
                    if (popc(ballot(cn)) * NODE_KEEP < popc(bn)) break;
                }
            }
        }
        };
        if (ballot(slow & active) != 0ull) streams(std::true_type{});
        else streams(std::false_type{});
        // hang guard: an active lane that can take neither stream (cannot happen while STACK > tree depth) ends here
        const bool stuck = active & !can_node & !can_tri & ((bn | bt) == 0ull);
        const bool done = active & (occluded | stuck | ((cur == PT_REF_NONE) & (lp == top)));
        if (done) { io.finish(slot, best, occluded); active = false; cur = PT_REF_NONE; lp = top; }
    }
#ifdef PT_UTIL_STATS
    if (lane == 0u) for (int i = 0; i < 8; i++) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
    for (int i = 8; i < 10; i++) if (ut[i]) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
#endif
}

// ---- PER-WAVE WORK LIST (ptmi_options.worklist) --------------------------------------------------------------------------
// The loop above lets every lane test the 1 - 4 triangles of ITS OWN leaf: a triangle iteration runs with 0.59 of the lanes
// (Cornell, profiles/r02_cfg1_lane_stats.json). Here a lane that opens a leaf only LISTS its triangles: (its lane number,
// triangle) items go into a ring of WL_RING words in LDS that belongs to the wave, and whenever the ring holds 64 items all 64
// lanes take one each — the ray (origin, direction; the any-hit limit) comes from the listing lane's registers through
// ds_bpermute, the triangle from the scene image as before. Results go back through LDS: the closest hit as ONE 64-bit minimum
// on (bits(t) << 32 | triangle) per ray — for t > 0 the order of the bit patterns is the order of the floats, so the minimum IS
// the contract's (smallest t, lowest triangle index) rule of pt.wgsl:274 / DESIGN.md §3.2 — the any-hit verdict as a flag.
// A ray's distance limit then lags by up to a ring's worth of triangles, which is conservative (a stale limit culls less).
// A ray is finished when it has no node, no filed leaf and no listed triangle left (my_end <= head); partly filled rounds of 64
// are only run when fewer than PT_WL_FLUSH_BELOW lanes have box work left (the others' rays wait for exactly those triangles).
// An occluded shadow ray finishes at once; its lane is refilled only after its listed triangles have drained (their owner lane
// must not change).
// The box stream is the one above, unchanged.
#ifndef PT_WL_RING
#define PT_WL_RING 256
#endif
#ifndef PT_WL_FLUSH_BELOW
#define PT_WL_FLUSH_BELOW 24        /* fewer lanes than this with box work left: test the listed triangles now, full round or not */
#endif
#ifndef PT_WL_LIST_MIN
#define PT_WL_LIST_MIN 12           /* a further listing pass only for at least this many lanes (while box work remains) */
#endif
constexpr uint32_t WL_RING = PT_WL_RING;                  // items (a power of two, >= 64 + PT_LEAF_MAX_TRIS)
constexpr uint32_t WL_WORDS = 128u + WL_RING;            // per wave: 64 keys of 8 bytes, then the ring
static_assert((WL_RING & (WL_RING - 1u)) == 0u && WL_RING >= 64u + PT_LEAF_MAX_TRIS, "ring size");
typedef __attribute__((address_space(3))) unsigned long long *lds_u64p;
PT_DEV uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
PT_DEV float bperm(uint32_t byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_addr, __float_as_int(v))); }

template <int MODE, bool CULL, int STACK, int REFILL, class Mem, class IO>
PT_DEV void trace_wave_wl(const Mem &m, const DevScene &sc, const IO &io, uint32_t count, uint32_t gw,
                          uint32_t total_waves, uint32_t *stk, int stride, uint32_t *wl) {
    constexpr bool ANY = MODE == MODE_SHADOW;
    constexpr int NODE_KEEP = ANY ? 2 : 3;
    constexpr unsigned long long KEY_NONE = ANY ? 0ull : ((0x7F800000ull << 32) | 0xFFFFFFFFull);     // (t = +inf, no triangle)
    const uint32_t lane = threadIdx.x & 63u;
    gw = uniform(gw);
    const uint32_t ngroups = (count + 63u) >> 6;
    const uint32_t end = gw < ngroups ? ((ngroups - gw + total_waves - 1u) / total_waves) * 64u : 0u;
    uint32_t next = 0u;
    bool active = false, slow = false, occ = false;
    const bool has_fast = sc.has_fast != 0u;
    const uint32_t leaf_bits = uniform(sc.leaf_bits);
    uint32_t slot = 0, cur = PT_REF_NONE;
    const lds_u32p bot = (lds_u32p)stk, top = bot + (STACK - 1) * stride;
    lds_u32p sp = bot, lp = top;
    const lds_u64p keys = (lds_u64p)wl;
    const lds_u32p ring = (lds_u32p)wl + 128;
    uint32_t head = 0u, tail = 0u;          // wave-uniform: items [head, tail) are listed and not yet tested (indices modulo WL_RING)
    uint32_t my_end = 0u;                   // tail after this lane's last listing: it has items pending while my_end - head > 0
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 0);
    float tlim = 0.0f, limit = __builtin_inff();
    keys[lane] = KEY_NONE;
#ifdef PT_UTIL_STATS
    uint32_t ut[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (;;) {
        bool pend = (int)(my_end - head) > 0;
        uint64_t act = ballot(active);
        if (next < end && popc(act) <= REFILL) {
            const bool free_lane = !active & !pend;
            const uint64_t idle = ballot(free_lane);
            const uint32_t vi = next + mbcnt(idle);
            const uint32_t vslot = ((vi >> 6) * total_waves + gw) * 64u + (vi & 63u);
            if (free_lane && vi < end && vslot < count) {
                slot = vslot;
                const bool want = io.fetch(slot, o, d, tlim);
                inv = mk3(rcp1(d.x), rcp1(d.y), rcp1(d.z));
                keys[lane] = KEY_NONE; occ = false;
                sp = bot; lp = top; cur = PT_REF_NONE; my_end = head;
                limit = (ANY && CULL) ? cull_limit(tlim) : __builtin_inff();
                const bool regular = __builtin_isfinite(inv.x) & __builtin_isfinite(inv.y) & __builtin_isfinite(inv.z) &
                                     (inv.x != 0.0f) & (inv.y != 0.0f) & (inv.z != 0.0f);
                const bool bounded = (__builtin_fabsf(d.x) + __builtin_fabsf(d.y) + __builtin_fabsf(d.z)) <= sc.tri_safe_dsum;
                slow = !(regular & bounded);
                float tm;
                if (want && sc.root_ref != PT_REF_NONE &&
                    slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2],
                         o, inv, tm)) {
                    active = true;
                    const uint32_t r = (has_fast & slow) ? sc.ref_root_ref : sc.root_ref;
                    if (r & PT_REF_LEAF) { *lp = r; lp -= stride; }
                    else cur = r;
                } else {
                    Hit none; none.t = __builtin_inff(); none.tri = PT_REF_NONE;
                    io.finish(slot, none, false);
                }
            }
            next += (uint32_t)popc(idle);
            UTIL(2, 1); UTIL(3, popc(ballot(active)) - popc(act));
            act = ballot(active);
        }
        if (act == 0ull) {
            if (next >= end) break;
            head = tail;                    // what is still listed belongs to finished (occluded) rays: drop it, their lanes are free again
            continue;
        }
        UTIL(0, 1); UTIL(1, popc(act));

        // One iteration = the three phases in a row, each skipped when it has nothing to do: box-pair steps for the lanes that
        // can take one; then every filed leaf is listed (a lane finds ~3 leaves in the 4.9 steps its ray takes, so after a
        // round of steps most lanes have some); then listed triangles are tested in rounds of 64, the rest too once few lanes
        // have box work left (their rays wait for exactly those triangles). (A first version voted for ONE phase per
        // iteration like the per-lane loop does: 7.3 votes per 64 rays instead of 2.7, and with them +31 % instructions.)
        const bool can_node = active & (cur != PT_REF_NONE) & room2(lp, sp, stride);
        const uint64_t bn = ballot(can_node);
        bool worked = false;
        auto streams = [&](auto with_ref) {
        constexpr bool REF = decltype(with_ref)::value;
        if (bn != 0ull) {
            worked = true;
            bool cn = can_node;
#pragma unroll
            for (int rep = 0; rep < NODE_STEPS; rep++) {
                UTIL(4, 1); UTIL(5, popc(ballot(cn)));
                if (cn) {
                    float tl, tr;
                    bool hl, hr;
                    uint32_t lref, rref;
                    Boxes nb;
                    if (REF && has_fast && slow) {
                        float4 a, b, c, r;
                        load_node((glb_f4p)sc.ref_wnodes + 4u * (size_t)cur, a, b, c, r);
                        boxes_of(a, b, c, r, nb);
                    } else {
                        m.node(cur, false, nb);
                    }
                    hl = slab(nb.lx0, nb.ly0, nb.lz0, nb.lx1, nb.ly1, nb.lz1, o, inv, tl);
                    hr = slab(nb.rx0, nb.ry0, nb.rz0, nb.rx1, nb.ry1, nb.rz1, o, inv, tr);
                    lref = nb.lref; rref = nb.rref;
                    if (CULL) { hl = hl & !(tl > limit); hr = hr & !(tr > limit); }
                    const bool ll = (lref & PT_REF_LEAF) != 0u, rl = (rref & PT_REF_LEAF) != 0u;
                    if (hl & ll) { *lp = lref; lp -= stride; }
                    if (hr & rl) { *lp = rref; lp -= stride; }
                    const bool il = hl & !ll, ir = hr & !rl;
                    const bool left_first = tl <= tr;
                    if (il & ir) { *sp = left_first ? rref : lref; sp += stride; cur = left_first ? lref : rref; }
                    else if (il) cur = lref;
                    else if (ir) cur = rref;
                    else if (sp != bot) { sp -= stride; cur = *sp; }
                    else cur = PT_REF_NONE;
                }
                if (rep + 1 < NODE_STEPS) {
                    cn = cn & (cur != PT_REF_NONE) & room2(lp, sp, stride);
                    if (popc(ballot(cn)) * NODE_KEEP < popc(bn)) break;
                }
            }
        }
        // list the triangles of the filed leaves
        {
            uint32_t room = WL_RING - (tail - head);
#pragma unroll 1
            for (int rep = 0; rep < STACK; rep++) {
                const bool ct = active & (lp != top) & !occ;
                const uint64_t bct = ballot(ct);
                if (bct == 0ull) break;
                if (rep > 0 && popc(bct) < PT_WL_LIST_MIN && bn != 0ull) break;      // a few stragglers' leaves can wait for the next round
                worked = true;
                UTIL(6, 1); UTIL(7, popc(bct));
                uint32_t first = 0u, cnt = 0u, cursor;
                if (ct) open_plain(*(lp + stride), first, cnt, cursor);
                uint32_t pre = 0u, tot = 0u;                        // exclusive prefix and total of cnt over the wave, bit by bit
                for (uint32_t b = 0; b < leaf_bits; b++) {
                    const uint64_t mb = ballot(((cnt >> b) & 1u) != 0u);
                    pre += mbcnt(mb) << b; tot += (uint32_t)popc(mb) << b;
                }
                const bool ok = ct & (pre + cnt <= room);          // a prefix of the listing lanes (pre ascends with the lane)
                const uint64_t bok = ballot(ok);
                if (bok == 0ull) break;
                if (ok) {
                    for (uint32_t k = 0; k < cnt; k++) ring[(tail + pre + k) & (WL_RING - 1u)] = (lane << PT_LEAF_OFF_BITS) | (first + k);
                    my_end = tail + pre + cnt;
                    lp += stride;
                }
                uint32_t pushed = tot;
                if (tot > room) pushed = (uint32_t)__builtin_amdgcn_readlane((int)(pre + cnt), 63 - __builtin_clzll(bok));
                tail += pushed; room -= pushed;
                if (room < 64u) break;                              // test some before listing more
            }
        }
        // test listed triangles: whole rounds of 64, and the rest when few lanes have box work left
        {
            const int box_lanes = popc(ballot(active & (cur != PT_REF_NONE)));
            const bool flush = box_lanes < PT_WL_FLUSH_BELOW;
            while (tail - head >= 64u || (flush && tail != head)) {
                worked = true;
                const uint32_t n = tail - head < 64u ? tail - head : 64u;
                const bool valid = lane < n;
                UTIL(8, lane == 0u ? 1 : 0); UTIL(9, valid ? 1 : 0);
                uint32_t item = ring[(head + lane) & (WL_RING - 1u)];
                item = valid ? item : 0u;                          // lane 0's ray against triangle 0, result unused
                const uint32_t owner = item >> PT_LEAF_OFF_BITS, ti = item & PT_LEAF_OFF_MASK;
                const uint32_t oa = owner << 2;
                const v3 ro = mk3(bperm(oa, o.x), bperm(oa, o.y), bperm(oa, o.z));
                const v3 rd = mk3(bperm(oa, d.x), bperm(oa, d.y), bperm(oa, d.z));
                float4 a, b, c;
                m.tri(ti, 0u, false, a, b, c);
                float u = 0.0f, v = 0.0f;
                const float t = tri_test_t<!REF>(xyz(a), xyz(b), xyz(c), ro, rd, u, v);
                const bool hit = valid & (t > 0.0f);
                if (ANY) {
                    const float rl = bperm(oa, tlim);
                    if (hit & !(t >= rl)) *(lds_u32p)(keys + owner) = 1u;
                } else if (hit) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | ti;
                    __hip_atomic_fetch_min(keys + owner, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                head += n;
            }
            // every ray picks up what the rounds found for it
            if (ANY) occ = *(lds_u32p)(keys + lane) != 0u;
            else if (CULL) limit = cull_limit(__uint_as_float((uint32_t)(keys[lane] >> 32)));
        }
        };
        if (ballot(slow & active) != 0ull) streams(std::true_type{});
        else streams(std::false_type{});
        const int what = worked ? 0 : -1;
        pend = (int)(my_end - head) > 0;
        // hang guard: nothing could run for anybody (cannot happen while STACK > tree depth): active lanes end with what they have
        const bool done = active & (occ | (what < 0) | ((cur == PT_REF_NONE) & (lp == top) & !pend));
        if (done) {
            Hit best; best.t = __builtin_inff(); best.tri = PT_REF_NONE;
            if (!ANY) { const unsigned long long k = keys[lane]; best.t = __uint_as_float((uint32_t)(k >> 32)); best.tri = (uint32_t)k; }
            io.finish(slot, best, occ);
            active = false; cur = PT_REF_NONE; lp = top;
        }
    }
#ifdef PT_UTIL_STATS
    if (lane == 0u) for (int i = 0; i < 8; i++) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
    for (int i = 8; i < 10; i++) if (ut[i]) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
#endif
}

// ------------------------------------------------------------------ global ----
constexpr int GBLOCK = 256;

#ifndef PT_GLOBAL_WAVES
#define PT_GLOBAL_WAVES 0          /* > 0: ask the register allocator for at least that many waves per SIMD */
#endif
#if PT_GLOBAL_WAVES > 0
#define PT_GLOBAL_ATTR __attribute__((amdgpu_waves_per_eu(PT_GLOBAL_WAVES)))
#else
#define PT_GLOBAL_ATTR
#endif
template <int MODE, bool CULL, int STACK, bool QUANT, class IO>
__global__ __launch_bounds__(GBLOCK) PT_GLOBAL_ATTR void k_trace_global(DevScene sc, IO io, const uint32_t *__restrict__ count_ptr,
                                                         uint32_t *__restrict__ spill, uint32_t *ticket) {
    __shared__ uint32_t stk[STACK * GBLOCK];
    const uint32_t count = *count_ptr;
    const uint32_t gw = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;       // consecutive groups -> different workgroups
    uint32_t *sp = spill + (size_t)blockIdx.x * GBLOCK + threadIdx.x;
    if constexpr (QUANT) {
        // the top of the tree (every ray's first steps) is read from LDS: the upload numbers those nodes breadth-first.
        // Filled by ALL 256 threads, before any wave may leave (a wave without rays would otherwise leave its share unfilled)
        __shared__ uint4 qcache[2 * PT_QCACHE_NODES];
        const uint32_t nc = sc.q_cached < PT_QCACHE_NODES ? sc.q_cached : PT_QCACHE_NODES;
        if (blockIdx.x * 64u >= count) return;                               // wave 0 owns the lowest group: the whole workgroup is idle
        for (uint32_t i = threadIdx.x; i < 2u * nc; i += GBLOCK) qcache[i] = sc.qnodes[i];
        __syncthreads();
        if (gw * 64u >= count) return;
        QuantMem m{(glb_u4p)sc.qnodes, (glb_u32p)sc.leaf_stream, (glb_f4p)sc.tripos,
                   sc.q_origin[0], sc.q_origin[1], sc.q_origin[2], sc.q_scale[0], sc.q_scale[1], sc.q_scale[2],
                   (lds_u4p)qcache, nc};
        trace_wave<MODE, CULL, STACK, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK, ticket);
    } else {
        if (gw * 64u >= count) return;
        GlobalMem m{(glb_f4p)sc.wnodes, (glb_f4p)sc.tripos};
        trace_wave<MODE, CULL, STACK, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK, ticket);
    }
}

// --------------------------------------------------------------------- LDS ----
// Persistent 1024-thread workgroups stage the traversal image into LDS once, then their 16 waves walk
// their share of the queue. Two footprints:
//   full       wide nodes + triangle images + stacks (Cornell: 20 + 47 + 64 KB): one workgroup per CU
//   node cache wide nodes + stacks only, triangle images through L1/L2: when that is <= 80 KB two
//              workgroups fit a CU (8 waves per SIMD instead of 4)
// Measured on Cornell 1080p: closest-hit rays run 14 % faster from the node cache (they are
// issue-bound and gain from the second workgroup), shadow rays 7 % slower (they test fewer boxes per
// triangle and miss the LDS-resident triangles); ptmi_api picks per kernel.
constexpr int LBLOCK = 1024;

#ifndef PT_LDS_WAVES
#define PT_LDS_WAVES 0             /* > 0: ask the register allocator for at least that many waves per SIMD (8 = two 1024-thread workgroups per CU) */
#endif
#if PT_LDS_WAVES > 0
#define PT_LDS_ATTR __attribute__((amdgpu_waves_per_eu(PT_LDS_WAVES)))
#else
#define PT_LDS_ATTR
#endif
template <int MODE, bool CULL, int STACK, bool TRIS_IN_LDS, bool SPILL, bool WL, class IO>
__global__ __launch_bounds__(LBLOCK) PT_LDS_ATTR void k_trace_lds(DevScene sc, IO io, const uint32_t *__restrict__ count_ptr,
                                                      uint32_t *__restrict__ spill, uint32_t *ticket) {
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * 64u >= count) return;      // wave 0 owns group blockIdx.x; if that is empty the whole group is idle
    const uint32_t nw = 4u * sc.n_wnodes, nt = TRIS_IN_LDS ? 3u * sc.n_tris : 0u;
    for (uint32_t i = threadIdx.x; i < nw; i += LBLOCK) smem[i] = sc.wnodes[i];
    for (uint32_t i = threadIdx.x; i < nt; i += LBLOCK) smem[nw + i] = sc.tripos[i];
    __syncthreads();
    const uint32_t gw = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    if (gw * 64u >= count) return;
    LdsMem<TRIS_IN_LDS> m{(lds_f4p)smem, (lds_f4p)(smem + nw), (glb_f4p)sc.tripos};
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem + nw + nt) + threadIdx.x;
    if constexpr (WL) {
        static_assert(!SPILL, "the work-list loop has no spilling stack");
        uint32_t *wl = reinterpret_cast<uint32_t *>(smem + nw + nt) + (size_t)STACK * LBLOCK + (threadIdx.x >> 6) * WL_WORDS;
        trace_wave_wl<MODE, CULL, STACK, REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk, LBLOCK, wl);
    } else {
        trace_wave<MODE, CULL, STACK, SPILL, REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk, LBLOCK,
                                             SPILL ? spill + (size_t)blockIdx.x * LBLOCK + threadIdx.x : nullptr, gridDim.x * LBLOCK, ticket);
    }
}

// the ticket counter of the launch being enqueued (PT_DYNAMIC_CLAIM builds; set by launch() from TraverseConfig::ticket)
thread_local uint32_t *g_ticket = nullptr;

template <int MODE, bool CULL, int STACK, bool TRIS, bool SPILL = false, bool WL = false, class IO>
void launch_lds(hipStream_t s, int wgs, size_t bytes, const DevScene &sc, const IO &io, const uint32_t *count,
                uint32_t *spill = nullptr) {
    // the default dynamic-LDS cap is 64 KB; raise it once per instantiation and device
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_lds<MODE, CULL, STACK, TRIS, SPILL, WL, IO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_trace_lds<MODE, CULL, STACK, TRIS, SPILL, WL, IO>), dim3(wgs), dim3(LBLOCK), bytes, s, sc, io, count, spill, g_ticket);
}

// The persistent grid of the global variant is exactly the workgroups that are resident at once: every workgroup
// carries a full share of the queue, so one more per CU than fit runs a second, almost empty round. Sweep (extend, ms
// per 64 spp, workgroups per CU; 16 LDS entries per lane + spill area):
//   cornell_spheres  3: 34.7  4: 30.7  5: 28.6  6: 25.9  7: 33.1  8: 30.8
//   grid_1m          3: 28.9  4: 24.9  5: 23.7  6: 23.1  7: 28.9  8: 27.1
// 6 is what the kernel's registers allow (4 waves per workgroup, 6 waves per SIMD); the occupancy query reports it.
// (Before the stacks could spill, depth-29 grid_1m needed 32 LDS entries per lane: 4 workgroups per CU, 23.8 ms.)
constexpr int GLOBAL_WGS_MAX = 8;          // what the spill area is sized for
template <int MODE, bool CULL, bool QUANT, class IO>
void launch_global_q(hipStream_t s, int cus, const DevScene &sc, const IO &io, const uint32_t *count, uint32_t *spill) {
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_global<MODE, CULL, 16, QUANT, IO>, GBLOCK, 0) != hipSuccess || n < 1) n = 6;
        per_cu = n < GLOBAL_WGS_MAX ? n : GLOBAL_WGS_MAX;
    }
    hipLaunchKernelGGL((k_trace_global<MODE, CULL, 16, QUANT, IO>), dim3(per_cu * cus), dim3(GBLOCK), 0, s, sc, io, count, spill, g_ticket);
}
template <int MODE, bool CULL, class IO>
void launch_global(hipStream_t s, int cus, const DevScene &sc, const IO &io, const uint32_t *count, uint32_t *spill, bool quant) {
    if (quant && sc.qnodes) launch_global_q<MODE, CULL, true>(s, cus, sc, io, count, spill);
    else launch_global_q<MODE, CULL, false>(s, cus, sc, io, count, spill);
}

template <int MODE, bool CULL, class IO>
void launch(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, const IO &io,
            const uint32_t *count) {
    const int cus = blocks / 8 > 0 ? blocks / 8 : 1;
    g_ticket = cfg.ticket;
    const size_t stack_bytes = (size_t)cfg.stack_entries * LBLOCK * sizeof(uint32_t);
    if (cfg.variant == PT_VARIANT_LDS_NODES && cfg.wgs_per_cu == 1) {
        // mid-size trees: all wide nodes in LDS next to 16 stack entries per lane (deeper stacks spill), one workgroup per CU
        const size_t bytes = (size_t)sc.n_wnodes * 64 + (size_t)16 * LBLOCK * sizeof(uint32_t);
        launch_lds<MODE, CULL, 16, false, true>(s, cus, bytes, sc, io, count, cfg.spill);
    } else if (cfg.variant == PT_VARIANT_LDS_NODES) {          // node cache, two workgroups per CU
        const size_t bytes = (size_t)sc.n_wnodes * 64 + stack_bytes;
#ifndef PT_SHADOW_NODE_CACHE_WGS
#define PT_SHADOW_NODE_CACHE_WGS 2
#endif
        const int wgs = (MODE == MODE_SHADOW ? PT_SHADOW_NODE_CACHE_WGS : 2) * cus;
        if (cfg.stack_entries <= 15) launch_lds<MODE, CULL, 15, false>(s, wgs, bytes, sc, io, count);
        else launch_lds<MODE, CULL, 16, false>(s, wgs, bytes, sc, io, count);
    } else if (cfg.variant == PT_VARIANT_LDS) {                // everything resident, one workgroup per CU
        const size_t bytes = cfg.lds_scene_bytes + stack_bytes;
        if (cfg.worklist && cfg.stack_entries <= 16) launch_lds<MODE, CULL, 16, true, false, true>(s, cus, bytes + pt_worklist_bytes(), sc, io, count);
        else if (cfg.stack_entries <= 16) launch_lds<MODE, CULL, 16, true>(s, cus, bytes, sc, io, count);
        else launch_lds<MODE, CULL, 32, true>(s, cus, bytes, sc, io, count);
    } else {
        launch_global<MODE, CULL>(s, cus, sc, io, count, cfg.spill, cfg.quantized != 0);
    }
}

}  // namespace

void pt_launch_extend(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      const uint32_t *queue, const uint32_t *count, float2 *hits) {
    ExtendIO io{p.O, p.D, queue, hits};
    if (cfg.cull) launch<MODE_EXTEND, true>(s, blocks, cfg, sc, io, count);
    else launch<MODE_EXTEND, false>(s, blocks, cfg, sc, io, count);
}

void pt_launch_shadow(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      DevShadow sh, const uint32_t *shadow_queue, const uint32_t *count, uint8_t *occ) {
    if (occ) {                                  // ptmi_debug_occluded (never with a queue)
        OccludedIO io{sh.SO, occ, sh.cap};
        if (cfg.cull) launch<MODE_SHADOW, true>(s, blocks, cfg, sc, io, count);
        else launch<MODE_SHADOW, false>(s, blocks, cfg, sc, io, count);
        return;
    }
    ShadowIO io{p.L, sh.SO, shadow_queue, p.l_stride, sh.cap};
    if (cfg.cull) launch<MODE_SHADOW, true>(s, blocks, cfg, sc, io, count);
    else launch<MODE_SHADOW, false>(s, blocks, cfg, sc, io, count);
}

int pt_dynamic_claim(void) { return PT_DYNAMIC_CLAIM; }

size_t pt_worklist_bytes(void) { return (size_t)(LBLOCK / 64) * WL_WORDS * sizeof(uint32_t); }

size_t pt_spill_bytes(int blocks) {
    const int cus = blocks / 8 > 0 ? blocks / 8 : 1;
    return (size_t)GLOBAL_WGS_MAX * cus * GBLOCK * PT_SPILL_ENTRIES * sizeof(uint32_t);
}

#ifdef PT_UTIL_STATS
extern "C" __attribute__((visibility("default"))) int ptmi_debug_util_stats(unsigned long long *out32, int reset) {
    unsigned long long h[32];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_util), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 32; i++) out32[i] = h[i];
    if (reset) { for (auto &x : h) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_util), h, sizeof(h)) != hipSuccess) return 1; }
    return 0;
}
#endif
