// traverse_own.hip — the traversal kernels over the library's OWN leaves (ptmi_options.leaves = 2, the default).
//
//   extend : closest hit per queued path  (reference: src/shader/pt.wgsl:248-296, Moller-Trumbore part of :123-158)
//   shadow : any-hit visibility of the next-event record written by `shade` (pt.wgsl:392/421/463, predicates :394/:423/:465)
//
// Why own leaves. The reference's builder (src/renderer/bvh.ts:86-127) cuts leaves of <= 4 triangles from 11 equal-count
// candidates on one axis; a Cornell ray crosses three unrelated 3-triangle leaves and tests ten triangles where 2.4 suffice, and
// the triangle tests are 60 - 70 % of the instructions of the kernels in traverse.hip, which keep exactly those leaves. pt.wgsl's
// result is the minimum t over whatever is tested (:270-280), so any leaves will do IF the same minimum comes out.
//
// Result contract. The reference tests triangle T iff the box of T's reference leaf passes the slab test
// (regular rays: the predicate is monotone under containment, so a passing leaf box implies passing ancestors). Here:
//   1. the descent runs over a SAH hierarchy of the triangles themselves (fast_tree.hip pt_build_own_tree) whose boxes are PADDED
//      so that the fused slab test fma(bound, 1/d, -o/d) — 12 instructions a box pair fewer than (bound - o) * (1/d) — accepts
//      every ray the exact box of the triangle would: the set of triangles tested is a superset of those the ray actually meets;
//   2. the WINNER (closest hit / the occluder that ends a shadow ray) is verified before it is reported: the box of its reference
//      leaf (per-triangle table DevScene::tri_leafbox) is tested with the contract's own slab arithmetic. If it passes, the winner
//      is a triangle the reference tests too, and being the minimum over a superset it is the reference's minimum;
//   3. if it fails — the reference would never have tested that triangle: a ray grazing the leaf box within rounding — the ray is
//      traced again over the tree exactly as uploaded (the `slow` path below), which is the reference's computation itself. The same
//      path takes every ray the argument of 1. does not cover: a zero / subnormal / non-finite direction component or one outside
//      [2^-60, 2^60], an origin farther than DevScene::safe_origin from the coordinate origin, an unbounded triangle-test determinant.
// So results equal traverse.hip's bit for bit unless a ray meets a triangle in Moller-Trumbore's arithmetic while missing that
// triangle's padded bounding box — the padding is 2^-16 of the scene's largest coordinate, 16 x the rounding of the fused test for
// origins inside the scene; tests/test_gpu_own_leaves.py counts such rays (none in 10^8 and more).
//
// Execution model, majority scheduling and deferred leaves are those of traverse.hip (see there); the memory variants:
//   OwnLdsMem     exact 64-byte nodes in LDS (+ the triangle images when they fit too)
//   OwnQuantMem   32-byte nodes — child boxes as 16-bit plane numbers, decoded and tested in ONE fma per plane
//                 (fma(k, scale / d, (origin - o) / d)) — all of them in LDS (small and mid-size scenes: two workgroups per CU, or
//                 one with every node of a 3 000-node tree resident), or the top of the tree in LDS and the rest through L1 / L2
//   OwnGlobalMem  exact nodes from memory (scenes a 16-bit grid cannot resolve; PTMI_TRAVERSAL_GLOBAL_EXACT)
#include "pt_device.h"
#include "pt_math.h"
#include "traverse_common.h"
#include <atomic>
#include <type_traits>

namespace {

// refill when at most this many of the 64 lanes still hold a ray (scene in LDS; traverse_common.h has the value of the other kernels)
#ifndef PT_OWN_REFILL_AT
#define PT_OWN_REFILL_AT 44   /* (36 until session 24: a refill is 35 vector instructions and seven loads cheaper now; config 2 +1 %, config 1 +-0) */
#endif

typedef const __attribute__((address_space(1))) uint32_t *glb_u32p;
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u4v *glb_u4p;
typedef const __attribute__((address_space(3))) u4v *lds_u4p;

// fused slab test of a conservative box: t = fma(bound, 1/d, -o/d) per plane
PT_DEV bool slab_fma(float bx0, float by0, float bz0, float bx1, float by1, float bz1, v3 inv, v3 n, float &tmin) {
    const float t1x = fma1(bx0, inv.x, n.x), t2x = fma1(bx1, inv.x, n.x);
    const float t1y = fma1(by0, inv.y, n.y), t2y = fma1(by1, inv.y, n.y);
    const float t1z = fma1(bz0, inv.z, n.z), t2z = fma1(bz1, inv.z, n.z);
    tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    const float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= tmin && tmax >= 0.0f;
}
PT_DEV bool slab_t(float t1x, float t2x, float t1y, float t2y, float t1z, float t2z, float &tmin) {
    tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    const float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= tmin && tmax >= 0.0f;
}
PT_DEV v3 rcp3(v3 d) { return mk3(rcp1(d.x), rcp1(d.y), rcp1(d.z)); }

// ---- node / triangle access policies ---------------------------------------------------------------------------------
// Pre            : what a ray keeps for the box tests (computed once per ray by prep)
// test(i, pre..) : both child boxes of node i against the ray: entry distances, verdicts, child references
// tri(i, a, b, c): (v0, original index), e1, e2 of the triangle at position i of the leaf-ordered image
template <bool TRIS_IN_LDS>
struct OwnLdsMem {
    lds_f4p wn, tl; glb_f4p tg;
    struct Pre { v3 inv, n; };
    PT_DEV Pre prep(v3 o, v3 inv) const { return Pre{inv, mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z))}; }
    PT_DEV v3 inv_of(const Pre &p, v3) const { return p.inv; }          // 1 / d as the ray keeps it
    PT_DEV void test(uint32_t i, const Pre &p, float &tl_, float &tr_, bool &hl, bool &hr, uint32_t &lref, uint32_t &rref) const {
        lds_f4p q = wn + 4u * i;
        const float4 a = as_f4(q[0]), b = as_f4(q[1]), c = as_f4(q[2]), r = as_f4(q[3]);
        hl = slab_fma(a.x, a.y, a.z, a.w, b.x, b.y, p.inv, p.n, tl_);
        hr = slab_fma(b.z, b.w, c.x, c.y, c.z, c.w, p.inv, p.n, tr_);
        lref = __float_as_uint(r.x); rref = __float_as_uint(r.y);
    }
    PT_DEV void tri(uint32_t i, float4 &a, float4 &b, float4 &c) const {
        if (TRIS_IN_LDS) { lds_f4p p = tl + 3u * i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
        else { glb_f4p p = tg + 3u * (size_t)i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
    }
};
struct OwnGlobalMem {
    glb_f4p wn, tg;
    struct Pre { v3 inv, n; };
    PT_DEV Pre prep(v3 o, v3 inv) const { return Pre{inv, mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z))}; }
    PT_DEV v3 inv_of(const Pre &p, v3) const { return p.inv; }
    PT_DEV void test(uint32_t i, const Pre &p, float &tl_, float &tr_, bool &hl, bool &hr, uint32_t &lref, uint32_t &rref) const {
        float4 a, b, c, r; load_node(wn + 4u * (size_t)i, a, b, c, r);
        hl = slab_fma(a.x, a.y, a.z, a.w, b.x, b.y, p.inv, p.n, tl_);
        hr = slab_fma(b.z, b.w, c.x, c.y, c.z, c.w, p.inv, p.n, tr_);
        lref = __float_as_uint(r.x); rref = __float_as_uint(r.y);
    }
    PT_DEV void tri(uint32_t i, float4 &a, float4 &b, float4 &c) const {
        glb_f4p p = tg + 3u * (size_t)i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]);
    }
};
// Quantised nodes (fast_tree.hip pt_quantize_nodes): per child three words of 16-bit plane numbers (lo.x | lo.y << 16,
// lo.z | hi.x << 16, hi.y | hi.z << 16) and its reference; plane k on axis a stands at origin[a] + k * scale[a], the padded box rounded
// OUTWARD to that grid. The test never forms that position: t = fma(k, scale / d, (origin - o) / d), one instruction per plane after
// the conversion of k. Its rounding (<= 2^-22 of scene extent + origin distance, in position terms) is far inside the padding.
template <bool ALL_IN_LDS, bool TRIS_IN_LDS>
struct OwnQuantMem {
    lds_u4p qc; glb_u4p qn; uint32_t n_cached; lds_f4p tl; glb_f4p tg;
    float ox, oy, oz, sx, sy, sz;
    struct Pre { v3 s, o; };
    PT_DEV Pre prep(v3 o, v3 inv) const {
        const v3 n = mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
        return Pre{mk3(sx * inv.x, sy * inv.y, sz * inv.z), mk3(fma1(ox, inv.x, n.x), fma1(oy, inv.y, n.y), fma1(oz, inv.z, n.z))};
    }
    PT_DEV v3 inv_of(const Pre &, v3 d) const { return rcp3(d); }        // (the quantised test keeps scale / d, not 1 / d)
    PT_DEV void test(uint32_t i, const Pre &p, float &tl_, float &tr_, bool &hl, bool &hr, uint32_t &lref, uint32_t &rref) const {
        u4v l, r;
        if (ALL_IN_LDS || i < n_cached) { l = qc[2u * i]; r = qc[2u * i + 1u]; }
        else { l = qn[2u * (size_t)i]; r = qn[2u * (size_t)i + 1u]; }
        hl = slab_t(fma1((float)(l.x & 0xFFFFu), p.s.x, p.o.x), fma1((float)(l.y >> 16), p.s.x, p.o.x),
                    fma1((float)(l.x >> 16), p.s.y, p.o.y), fma1((float)(l.z & 0xFFFFu), p.s.y, p.o.y),
                    fma1((float)(l.y & 0xFFFFu), p.s.z, p.o.z), fma1((float)(l.z >> 16), p.s.z, p.o.z), tl_);
        hr = slab_t(fma1((float)(r.x & 0xFFFFu), p.s.x, p.o.x), fma1((float)(r.y >> 16), p.s.x, p.o.x),
                    fma1((float)(r.x >> 16), p.s.y, p.o.y), fma1((float)(r.z & 0xFFFFu), p.s.y, p.o.y),
                    fma1((float)(r.y & 0xFFFFu), p.s.z, p.o.z), fma1((float)(r.z >> 16), p.s.z, p.o.z), tr_);
        lref = l.w; rref = r.w;
    }
    PT_DEV void tri(uint32_t i, float4 &a, float4 &b, float4 &c) const {
        if (TRIS_IN_LDS) { lds_f4p p = tl + 3u * i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
        else { glb_f4p p = tg + 3u * (size_t)i; a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]); }
    }
};

// One wave traces the 64-slot groups gw, gw + total_waves, ... of a queue of `count` slots, 64 rays in flight, refilled when
// REFILL or fewer lanes still hold one (traverse.hip explains the scheduling; this is the same loop over the own image).
// `slow` lanes walk the tree exactly as uploaded (DevScene::ref_wnodes, ref_tripos, both from global memory) with the contract's slab
// arithmetic — the reference's computation; a wave runs the copy of the streams that can do so only while it holds such a lane.
// E = uint16_t: COMPACT REFERENCES (scenes of up to 4 096 triangles): the per-lane entries are 16 bits — an internal node's index, or 0x8000 |
// (count - 1) << 12 | first triangle for a leaf — as the node images DevScene::wnodes16 / ref_wnodes16 carry them. Fifteen entries of a
// 1024-thread workgroup are then 30 KB instead of 60, and a tree of up to 780 EXACT nodes (59 vector instructions a box step instead of
// the quantised nodes' 66) runs two workgroups per CU.
// STACK = 0: `nstack` entries per lane, known at the launch only (the spill area holds 32-bit words: a 16-bit entry is widened there).
template <int MODE, bool CULL, int STACK, bool SPILL, int REFILL, class Mem, class IO, class E>
PT_DEV void trace_wave_own(const Mem &m, const DevScene &sc, const IO &io, uint32_t count, uint32_t gw,
                           uint32_t total_waves, E *stk, int stride, uint32_t *spill = nullptr, uint32_t spill_lanes = 0, int nstack = 0) {
    constexpr bool R16 = sizeof(E) == 2;
    static_assert(STACK != 0 || SPILL, "a stack sized at the launch may be shorter than the tree is deep");
    constexpr uint32_t LEAF_BIT = R16 ? 0x8000u : PT_REF_LEAF;
    typedef __attribute__((address_space(3))) E *lds_ep;
    auto room = [](lds_ep a, lds_ep b, int st) { return (int)((uint32_t)(uintptr_t)a - (uint32_t)(uintptr_t)b) >= st * (int)sizeof(E); };
    auto open_leaf = [](uint32_t ref, uint32_t &first, uint32_t &cnt) {
        if (R16) { first = ref & 0xFFFu; cnt = ((ref >> 12) & 7u) + 1u; }
        else { first = ref & PT_LEAF_OFF_MASK; cnt = ((ref >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u; }
    };
    constexpr bool ANY = MODE == MODE_SHADOW;
    // how long a stream keeps running after a vote (traverse_common.h has the meaning; the values here are measured on the own image:
    // 7 - 8 box steps and 1.4 - 2 short leaves per ray instead of 5 and 3 longer ones)
#ifndef PT_OWN_NODE_STEPS
#define PT_OWN_NODE_STEPS PT_NODE_STEPS
#endif
#ifndef PT_OWN_LEAF_STEPS
#define PT_OWN_LEAF_STEPS PT_LEAF_STEPS
#endif
#ifndef PT_OWN_LEAF_KEEP
#define PT_OWN_LEAF_KEEP PT_LEAF_KEEP
#endif
    // The box stream goes on while at least 1 / NODE_KEEP of the lanes that started it can: with the scene in LDS longer than over the
    // reference's leaves (1/6 and 1/4 instead of 1/3 and 1/2: config 1 +1.5 ... +2 %, config 2 +2.3 %, tools/sessions/r04/s07.sh) — a ray is
    // mostly box steps now; the kernels that walk memory keep 1/3 and 1/2 (config 3: 6 / 4 is -1.5 %).
#ifndef PT_OWN_NODE_KEEP_EXTEND
#define PT_OWN_NODE_KEEP_EXTEND 6
#endif
#ifndef PT_OWN_NODE_KEEP_SHADOW
#define PT_OWN_NODE_KEEP_SHADOW 4
#endif
    constexpr bool FROM_MEMORY = REFILL == PT_REFILL_GLOBAL && PT_REFILL_GLOBAL != PT_OWN_REFILL_AT;
    constexpr int NODE_KEEP = FROM_MEMORY ? (ANY ? 2 : 3) : (ANY ? PT_OWN_NODE_KEEP_SHADOW : PT_OWN_NODE_KEEP_EXTEND);
    constexpr int NODE_STEPS = PT_OWN_NODE_STEPS, LEAF_STEPS = PT_OWN_LEAF_STEPS, LEAF_KEEP = PT_OWN_LEAF_KEEP;
    const uint32_t lane = threadIdx.x & 63u;
    gw = uniform(gw);
    const uint32_t ngroups = (count + 63u) >> 6;
    const uint32_t end = gw < ngroups ? ((ngroups - gw + total_waves - 1u) / total_waves) * 64u : 0u;
    uint32_t next = 0u;
    bool active = false, slow = false;
    bool fin = false, fin_occ = false;      // the ray has finished; its winner is verified and its result written at the next refill
    uint32_t slot = 0, cur = PT_REF_NONE, aux = 0;      // aux: what the ray's IO keeps from fetch to finish (ShadowIO: the path id)
    const lds_ep bot = (lds_ep)stk, top = bot + ((STACK ? STACK : nstack) - 1) * stride;
    lds_ep sp = bot, lp = top;
    uint32_t spn = 0;
    v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    typename Mem::Pre pre = m.prep(o, mk3(0, 0, 1));
    float tlim = 0.0f, limit = __builtin_inff();
    Hit best; best.t = __builtin_inff(); best.tri = PT_REF_NONE;       // ANY: best.tri = the occluder
    uint32_t n_redo = 0;
#ifdef PT_UTIL_STATS
    uint32_t ut[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // (re)start this lane's ray: the root test, then the root of the tree its kind of ray walks
    auto start = [&](bool slow_ray, v3 inv) -> bool {
        // (the boxes are uniform — scalar loads — and almost no wave holds a slow ray: its box is tested in a branch the whole wave
        // takes or skips; written as two operands selected per lane the compiler loads every bound from a per-lane address, six vector
        // loads and their latency in the wave's refill)
        float tm;
        bool hit = slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], o, inv, tm);
        uint32_t r = R16 ? sc.root_ref16 : sc.root_ref;
        if (ballot(slow_ray) != 0ull) {
            const bool h2 = slab(sc.ref_root_min[0], sc.ref_root_min[1], sc.ref_root_min[2], sc.ref_root_max[0], sc.ref_root_max[1], sc.ref_root_max[2], o, inv, tm);
            const uint32_t r2 = R16 ? sc.ref_root_ref16 : sc.ref_root_ref;
            hit = slow_ray ? h2 : hit; r = slow_ray ? r2 : r;
        }
        best.t = __builtin_inff(); best.tri = PT_REF_NONE;
        sp = bot; lp = top; spn = 0u; cur = PT_REF_NONE;
        limit = (ANY && CULL) ? cull_limit(tlim) : __builtin_inff();      // NaN for a directional light: never culls
        if (!hit) return false;
        if (r >= LEAF_BIT) { *lp = (E)r; lp -= stride; }                  // a one-leaf tree: file the root
        else cur = r;
        return true;
    };

    for (;;) {
        uint64_t act = ballot(active);
        // Finished rays wait for the wave's next refill (or its end) to be verified and written: the verification is a box test and
        // two fetches the whole wave steps through, and lanes finish in almost every iteration — once per refill instead of once per
        // iteration (2.8 -> 1.3 times per 64 rays on Cornell).
        if ((next < end && popc(act) <= REFILL) || act == 0ull) {
            if (fin) {
                // the winner must be a triangle the reference tests too: its reference leaf's box, the contract's slab test
                bool redo = false;
                if (!slow & (best.tri != PT_REF_NONE)) {
                    glb_f4p lb = (glb_f4p)sc.tri_leafbox + 2u * (size_t)best.tri;
                    const float4 lo = as_f4(lb[0]), hi = as_f4(lb[1]);
                    float tm;
                    redo = !slab(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, o, m.inv_of(pre, d), tm);
                }
                fin = false;
                if (redo) {                                             // never its own: the uploaded tree decides (at most once per ray)
                    slow = true; n_redo++;
                    if (start(true, rcp3(d))) active = true;
                    else io.finish(slot, best, false, aux);
                } else {
                    io.finish(slot, best, fin_occ, aux);
                }
            }
            act = ballot(active);
        }
        if (next < end && popc(act) <= REFILL) {
            const uint64_t idle = ~act;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t vi = next + rank;
            const uint32_t vslot = ((vi >> 6) * total_waves + gw) * 64u + (vi & 63u);
            if (!active && vi < end && vslot < count) {
                slot = vslot;
                const bool want = io.fetch(slot, o, d, tlim, aux);
                // the padding covers the fused test's rounding only for rays without huge or vanishing slopes, from origins near the
                // scene; every other ray is the reference's own business (NaN compares false: slow). A regular ray's direction
                // components lie within [2^-60, 2^60], where the short reciprocal is the correctly rounded one (pt_math.h) and needs no
                // range test of its own; the others get the IEEE quotient, in a branch a wave almost never takes.
                const float lo = 0x1p-60f, hi = 0x1p60f;
                const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
                const bool regular = (ax >= lo) & (ax <= hi) & (ay >= lo) & (ay <= hi) & (az >= lo) & (az <= hi);
                const bool bounded = (ax + ay + az) <= sc.tri_safe_dsum;
                const bool near_o = (__builtin_fabsf(o.x) <= sc.safe_origin) & (__builtin_fabsf(o.y) <= sc.safe_origin) &
                                    (__builtin_fabsf(o.z) <= sc.safe_origin);
                slow = !(regular & bounded & near_o);
                v3 inv = PT_IEEE_EXPANSIONS ? rcp3(d) : mk3(rcp_short(d.x), rcp_short(d.y), rcp_short(d.z));
                if (!PT_IEEE_EXPANSIONS && slow) inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                pre = m.prep(o, inv);
                if (want && sc.root_ref != PT_REF_NONE && start(slow, inv)) active = true;
                else { best.t = __builtin_inff(); best.tri = PT_REF_NONE; io.finish(slot, best, false, aux); }
            }
            next += (uint32_t)__popcll(idle);
            UTIL(2, 1); UTIL(3, popc(ballot(active)) - popc(act));
            act = ballot(active);
        }
        if (act == 0ull && next >= end) break;
        UTIL(0, 1); UTIL(1, popc(act));

        const bool can_node = active & (cur != PT_REF_NONE) & ((int)room(lp, sp, stride) | (int)(SPILL && (sp != bot) & room(lp, bot, stride)));
        const bool can_tri = active & (lp != top);
        const uint64_t bn = ballot(can_node), bt = ballot(can_tri);
        const bool run_tri = popc(bt) > popc(bn);
        bool occluded = false;
        auto streams = [&](auto with_ref) {
        constexpr bool REF = decltype(with_ref)::value;
        const bool old = REF && slow;
        v3 inv_old = mk3(0, 0, 0);
        if (REF) inv_old = rcp3(d);                                     // (the fast lanes keep what their box test needs in `pre`)
        if (run_tri) {
            bool ct = can_tri;
            uint64_t ctm = bt;
#pragma unroll 1
            for (int rep = 0; rep < LEAF_STEPS; rep++) {
                UTIL(6, 1); UTIL(7, popc(ballot(ct)));
                if (ct) {
                    lp += stride;                                       // next filed leaf
                    uint32_t first, cnt;
                    open_leaf(*lp, first, cnt);
                    for (uint32_t k = 0; k < cnt; k++) {                // pt.wgsl:272-279
                        UTIL(8, uniform(lane) == lane ? 1 : 0); UTIL(9, 1);
                        float4 a, b, c;
                        if (old) {
                            glb_f4p p = (glb_f4p)sc.ref_tripos + 3u * (size_t)(first + k);
                            a = as_f4(p[0]); b = as_f4(p[1]); c = as_f4(p[2]);
                        } else {
                            m.tri(first + k, a, b, c);
                        }
                        float u = 0.0f, v = 0.0f;
                        const float t = tri_test_t<!REF>(xyz(a), xyz(b), xyz(c), o, d, u, v);
                        const bool hit = t > 0.0f;
                        const uint32_t ti = old ? first + k : __float_as_uint(a.w);      // the ORIGINAL index either way
                        if (ANY) {
                            const bool occ = hit & !(t >= tlim);
                            best.tri = (occ & !occluded) ? ti : best.tri;
                            occluded = occluded | occ;
                        } else {
                            const bool better = hit & ((t < best.t) | ((t == best.t) & (ti < best.tri)));
                            best.t = better ? t : best.t; best.tri = better ? ti : best.tri;
                            if (CULL) limit = better ? cull_limit(t) : limit;
                        }
                    }
                }
                if (rep + 1 < LEAF_STEPS) {
                    // (the count comes from the lane masks of the comparisons themselves: a ballot of the combined verdict would first
                    // be rebuilt as a 0 / 1 integer per lane — two vector instructions a step)
                    ct = ct & (lp != top) & !occluded;
                    ctm = ctm & ballot(lp != top) & ~ballot(occluded);
                    if (popc(ctm) * LEAF_KEEP < popc(bt)) break;
                }
            }
        } else {
            bool cn = can_node;
            uint64_t cnm = bn;
            constexpr int NODE_UNROLL = SPILL ? PT_SPILL_NODE_UNROLL : NODE_STEPS;
#pragma unroll NODE_UNROLL
            for (int rep = 0; rep < NODE_STEPS; rep++) {
                UTIL(4, 1); UTIL(5, popc(ballot(cn)));
                if (cn) {
                    if (SPILL && !room(lp, sp, stride)) {               // rare: move the LDS node stack out
                        for (lds_ep q = bot; q != sp; q += stride) { spill[(size_t)spn * spill_lanes] = *q; spn++; }
                        sp = bot;
                    }
                    float tl, tr;
                    bool hl, hr;
                    uint32_t lref, rref;
                    if (old) {
                        float4 a, b, c, r;
                        load_node((glb_f4p)(R16 ? sc.ref_wnodes16 : sc.ref_wnodes) + 4u * (size_t)cur, a, b, c, r);
                        hl = slab(a.x, a.y, a.z, a.w, b.x, b.y, o, inv_old, tl);
                        hr = slab(b.z, b.w, c.x, c.y, c.z, c.w, o, inv_old, tr);
                        lref = __float_as_uint(r.x); rref = __float_as_uint(r.y);
                    } else {
                        m.test(cur, pre, tl, tr, hl, hr, lref, rref);
                    }
                    if (CULL) { hl = hl & !(tl > limit); hr = hr & !(tr > limit); }
                    // (a reference is a leaf iff its top bit is set — of the 32 bits, or of the 16 the compact references use: written as a
                    // comparison, which is one v_cmp into a lane mask; as a bit test the compiler builds the verdicts as 0 / 1 integers
                    // in vector registers, five instructions a child)
                    const bool ll = lref >= LEAF_BIT, rl = rref >= LEAF_BIT;
                    if (hl & ll) { *lp = (E)lref; lp -= stride; }
                    if (hr & rl) { *lp = (E)rref; lp -= stride; }
                    const bool il = hl & !ll, ir = hr & !rl;
                    const bool left_first = tl <= tr;
                    if (il & ir) { *sp = (E)(left_first ? rref : lref); sp += stride; cur = left_first ? lref : rref; }
                    else if (il) cur = lref;
                    else if (ir) cur = rref;
                    else if (sp != bot) { sp -= stride; cur = *sp; }
                    else if (SPILL && spn != 0u) {                      // rare: take the last 8 spilled entries back
                        const int fit = (int)((uint32_t)(uintptr_t)lp - (uint32_t)(uintptr_t)bot) / (stride * (int)sizeof(E));   // free - 1
                        uint32_t n = spn < 8u ? spn : 8u;
                        n = (int)n < fit ? n : (fit > 1 ? (uint32_t)fit : 1u);
                        spn -= n;
                        for (uint32_t j = 0; j + 1u < n; j++) { *sp = (E)spill[(size_t)(spn + j) * spill_lanes]; sp += stride; }
                        cur = spill[(size_t)(spn + n - 1u) * spill_lanes];
                    }
                    else cur = PT_REF_NONE;
                }
                if (rep + 1 < NODE_STEPS) {
                    cn = cn & (cur != PT_REF_NONE) & ((int)room(lp, sp, stride) | (int)(SPILL && (sp != bot) & room(lp, bot, stride)));
                    uint64_t rm = ballot(room(lp, sp, stride));
                    if (SPILL) rm |= ballot(sp != bot) & ballot(room(lp, bot, stride));
                    cnm = cnm & ballot(cur != PT_REF_NONE) & rm;
                    if (popc(cnm) * NODE_KEEP < popc(bn)) break;
                }
            }
        }
        };
        if (ballot(slow & active) != 0ull) streams(std::true_type{});
        else streams(std::false_type{});
        // hang guard: an active lane that can take neither stream (cannot happen while STACK > tree depth) ends here
        const bool stuck = active & !can_node & !can_tri & ((bn | bt) == 0ull);
        const bool done = active & (occluded | stuck | ((cur == PT_REF_NONE) & (lp == top)));
        if (done) {
            // (a stuck lane reports what it has, unverified: it cannot happen while STACK > tree depth)
            active = false; cur = PT_REF_NONE; lp = top;
            fin = true; fin_occ = occluded;
            if (stuck) slow = true;
        }
    }
    if (n_redo) atomicAdd(sc.verify_stat, (unsigned long long)n_redo);
#ifdef PT_UTIL_STATS
    if (lane == 0u) for (int i = 0; i < 8; i++) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
    for (int i = 8; i < 10; i++) if (ut[i]) atomicAdd(&g_util[MODE][i], (unsigned long long)ut[i]);
#endif
}

// ------------------------------------------------------------------ kernels ----
constexpr int GBLOCK = 256, LBLOCK = 1024;

#ifndef PT_OWN_LDS_WAVES
#define PT_OWN_LDS_WAVES 0
#endif
#if PT_OWN_LDS_WAVES > 0
#define PT_OWN_LDS_ATTR __attribute__((amdgpu_waves_per_eu(PT_OWN_LDS_WAVES)))
#else
#define PT_OWN_LDS_ATTR
#endif

// LAYOUT: 0 exact nodes (64 B) in LDS, 1 quantised nodes (32 B) in LDS, 2 exact nodes with compact references and 16-bit entries,
// 3 quantised nodes with compact references and `nstack` 16-bit entries (STACK = 0, SPILL).
// TRIS: the triangle images in LDS too. Dynamic LDS: [nodes][triangles][STACK x 1024 entries]
// (STACK = 15 is the footprint of two workgroups per CU: 8 waves per SIMD, which the register allocator has to be told — at most 64
// vector registers; left alone the max-ILP scheduler takes 72)
template <int MODE, bool CULL, int STACK, int LAYOUT, bool TRIS, bool SPILL, class IO>
__global__ __launch_bounds__(LBLOCK) __attribute__((amdgpu_waves_per_eu((STACK == 15 || LAYOUT == 3) ? 8 : 4))) PT_OWN_LDS_ATTR void k_own_lds(const DevScene *__restrict__ scp, IO io, const uint32_t *__restrict__ count_ptr,
                                                                    uint32_t *__restrict__ spill, int nstack) {
    // The scene description is read from memory where it is needed (the root boxes and limits at a refill, the uploaded tree by slow
    // rays, the leaf-box table at a verification) instead of living in scalar registers for the whole kernel: passed by value the
    // kernel took 104 of them, and two 1024-thread workgroups share a CU only up to 80 (traverse.hip: ShadowIO).
    const DevScene &sc = *scp;
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * 64u >= count) return;      // wave 0 owns group blockIdx.x; if that is empty the whole group is idle
    const uint32_t nw = ((LAYOUT == 1 || LAYOUT == 3) ? 2u : 4u) * sc.n_wnodes, nt = TRIS ? 3u * sc.n_own_tris : 0u;
    const float4 *src = LAYOUT == 1 ? reinterpret_cast<const float4 *>(sc.qnodes) : LAYOUT == 3 ? reinterpret_cast<const float4 *>(sc.qnodes16)
                      : LAYOUT == 2 ? sc.wnodes16 : sc.wnodes;
    for (uint32_t i = threadIdx.x; i < nw; i += LBLOCK) smem[i] = src[i];
    for (uint32_t i = threadIdx.x; i < nt; i += LBLOCK) smem[nw + i] = sc.tripos[i];
    __syncthreads();
    const uint32_t gw = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    if (gw * 64u >= count) return;
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem + nw + nt) + threadIdx.x;
    uint32_t *sp = SPILL ? spill + (size_t)blockIdx.x * LBLOCK + threadIdx.x : nullptr;
    if constexpr (LAYOUT == 2) {
        OwnLdsMem<TRIS> m{(lds_f4p)smem, (lds_f4p)(smem + nw), (glb_f4p)sc.tripos};
        uint16_t *stk16 = reinterpret_cast<uint16_t *>(smem + nw + nt) + threadIdx.x;
        trace_wave_own<MODE, CULL, STACK, false, PT_OWN_REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk16, LBLOCK);
    } else if constexpr (LAYOUT == 3) {
        OwnQuantMem<true, false> m{(lds_u4p)smem, (glb_u4p)sc.qnodes16, sc.n_wnodes, (lds_f4p)nullptr, (glb_f4p)sc.tripos,
                                   sc.q_origin[0], sc.q_origin[1], sc.q_origin[2], sc.q_scale[0], sc.q_scale[1], sc.q_scale[2]};
        uint16_t *stk16 = reinterpret_cast<uint16_t *>(smem + nw + nt) + threadIdx.x;
        trace_wave_own<MODE, CULL, 0, true, PT_OWN_REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk16, LBLOCK, sp, gridDim.x * LBLOCK, nstack);
    } else if constexpr (LAYOUT == 1) {
        OwnQuantMem<true, TRIS> m{(lds_u4p)smem, (glb_u4p)sc.qnodes, sc.n_wnodes, (lds_f4p)(smem + nw), (glb_f4p)sc.tripos,
                                  sc.q_origin[0], sc.q_origin[1], sc.q_origin[2], sc.q_scale[0], sc.q_scale[1], sc.q_scale[2]};
        trace_wave_own<MODE, CULL, STACK, SPILL, PT_OWN_REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk, LBLOCK, sp, gridDim.x * LBLOCK);
    } else {
        OwnLdsMem<TRIS> m{(lds_f4p)smem, (lds_f4p)(smem + nw), (glb_f4p)sc.tripos};
        trace_wave_own<MODE, CULL, STACK, SPILL, PT_OWN_REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk, LBLOCK, sp, gridDim.x * LBLOCK);
    }
}

template <int MODE, bool CULL, int STACK, int LAYOUT, bool TRIS, bool SPILL, class IO>
void launch_own_lds(hipStream_t s, int wgs, size_t bytes, const DevScene *sc, const IO &io, const uint32_t *count, uint32_t *spill, int nstack = 0) {
    static std::atomic<uint64_t> raised{0};      // the default dynamic-LDS cap is 64 KB; raise it once per instantiation and device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_own_lds<MODE, CULL, STACK, LAYOUT, TRIS, SPILL, IO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_own_lds<MODE, CULL, STACK, LAYOUT, TRIS, SPILL, IO>), dim3(wgs), dim3(LBLOCK), bytes, s, sc, io, count, spill, nstack);
}

// from global memory: 256-thread workgroups, 16 LDS entries per lane + the spill area; QUANT: quantised nodes, the top of the tree in LDS
template <int MODE, bool CULL, bool QUANT, class IO>
__global__ __launch_bounds__(GBLOCK) void k_own_global(const DevScene *__restrict__ scp, IO io, const uint32_t *__restrict__ count_ptr, uint32_t *__restrict__ spill) {
    const DevScene &sc = *scp;
    __shared__ uint32_t stk[16 * GBLOCK];
    const uint32_t count = *count_ptr;
    const uint32_t gw = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    uint32_t *sp = spill + (size_t)blockIdx.x * GBLOCK + threadIdx.x;
    if constexpr (QUANT) {
        __shared__ uint4 qcache[2 * PT_QCACHE_NODES];
        const uint32_t nc = sc.q_cached < PT_QCACHE_NODES ? sc.q_cached : PT_QCACHE_NODES;
        if (blockIdx.x * 64u >= count) return;
        for (uint32_t i = threadIdx.x; i < 2u * nc; i += GBLOCK) qcache[i] = sc.qnodes[i];
        __syncthreads();
        if (gw * 64u >= count) return;
        OwnQuantMem<false, false> m{(lds_u4p)qcache, (glb_u4p)sc.qnodes, nc, (lds_f4p)nullptr, (glb_f4p)sc.tripos,
                                    sc.q_origin[0], sc.q_origin[1], sc.q_origin[2], sc.q_scale[0], sc.q_scale[1], sc.q_scale[2]};
        trace_wave_own<MODE, CULL, 16, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK);
    } else {
        if (gw * 64u >= count) return;
        OwnGlobalMem m{(glb_f4p)sc.wnodes, (glb_f4p)sc.tripos};
        trace_wave_own<MODE, CULL, 16, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK);
    }
}
constexpr int GLOBAL_WGS_MAX = 8;          // what the spill area is sized for (traverse.hip pt_spill_bytes)
template <int MODE, bool CULL, bool QUANT, class IO>
void launch_own_global(hipStream_t s, int cus, const DevScene *sc, const IO &io, const uint32_t *count, uint32_t *spill) {
    static int per_cu = 0;                   // the persistent grid is exactly the workgroups that are resident at once (traverse.hip)
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_own_global<MODE, CULL, QUANT, IO>, GBLOCK, 0) != hipSuccess || n < 1) n = 6;
        per_cu = n < GLOBAL_WGS_MAX ? n : GLOBAL_WGS_MAX;
    }
    hipLaunchKernelGGL((k_own_global<MODE, CULL, QUANT, IO>), dim3(per_cu * cus), dim3(GBLOCK), 0, s, sc, io, count, spill);
}

template <int MODE, bool CULL, class IO>
void launch_own(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &hsc, const IO &io, const uint32_t *count) {
    const int cus = blocks / 8 > 0 ? blocks / 8 : 1;
    const DevScene *sc = hsc.self;              // the kernels read the description from device memory
    const size_t node_bytes = (size_t)hsc.n_wnodes * (cfg.variant == PT_VARIANT_OWN_LDS || cfg.variant == PT_VARIANT_OWN_LDS_NODES || cfg.variant == PT_VARIANT_OWN_LDS16_NODES ? 64 : 32);
    const size_t tri_bytes = (size_t)hsc.n_own_tris * 48;
    const size_t stack_bytes = (size_t)cfg.stack_entries * LBLOCK * sizeof(uint32_t);
    switch (cfg.variant) {
    case PT_VARIANT_OWN_LDS:                    // exact nodes + triangles resident, one workgroup per CU
        if (cfg.stack_entries <= 16) launch_own_lds<MODE, CULL, 16, 0, true, false>(s, cus, node_bytes + tri_bytes + stack_bytes, sc, io, count, nullptr);
        else launch_own_lds<MODE, CULL, 32, 0, true, false>(s, cus, node_bytes + tri_bytes + stack_bytes, sc, io, count, nullptr);
        break;
    case PT_VARIANT_OWN_LDS_NODES:              // exact nodes resident, triangles through L1 / L2
        if (cfg.wgs_per_cu == 2) launch_own_lds<MODE, CULL, 15, 0, false, false>(s, 2 * cus, node_bytes + stack_bytes, sc, io, count, nullptr);
        else launch_own_lds<MODE, CULL, 16, 0, false, true>(s, cus, node_bytes + stack_bytes, sc, io, count, cfg.spill);
        break;
    case PT_VARIANT_OWN_QLDS:                   // quantised nodes + triangles resident, one workgroup per CU
        if (cfg.stack_entries <= 16) launch_own_lds<MODE, CULL, 16, 1, true, false>(s, cus, node_bytes + tri_bytes + stack_bytes, sc, io, count, nullptr);
        else launch_own_lds<MODE, CULL, 32, 1, true, false>(s, cus, node_bytes + tri_bytes + stack_bytes, sc, io, count, nullptr);
        break;
    case PT_VARIANT_OWN_QLDS_NODES:             // quantised nodes resident, triangles through L1 / L2
        if (cfg.wgs_per_cu == 2) launch_own_lds<MODE, CULL, 15, 1, false, false>(s, 2 * cus, node_bytes + stack_bytes, sc, io, count, nullptr);
        else launch_own_lds<MODE, CULL, 16, 1, false, true>(s, cus, node_bytes + stack_bytes, sc, io, count, cfg.spill);
        break;
    case PT_VARIANT_OWN_LDS16_NODES:            // exact nodes with compact references, 16-bit entries: two workgroups per CU
        launch_own_lds<MODE, CULL, 15, 2, false, false>(s, 2 * cus, node_bytes + stack_bytes / 2, sc, io, count, nullptr);
        break;
    case PT_VARIANT_OWN_QLDS16_NODES:           // quantised nodes with compact references: two workgroups per CU, stack_entries 16-bit entries, spills
        launch_own_lds<MODE, CULL, 0, 3, false, true>(s, 2 * cus, node_bytes + (size_t)cfg.stack_entries * LBLOCK * sizeof(uint16_t), sc, io, count, cfg.spill, cfg.stack_entries);
        break;
    case PT_VARIANT_OWN_QGLOBAL: launch_own_global<MODE, CULL, true>(s, cus, sc, io, count, cfg.spill); break;
    default: launch_own_global<MODE, CULL, false>(s, cus, sc, io, count, cfg.spill); break;
    }
}

}  // namespace

void pt_launch_extend_own(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                          const uint32_t *queue, const uint32_t *count, float2 *hits) {
    ExtendIO io{p.O, p.D, queue, hits};
    if (cfg.cull) launch_own<MODE_EXTEND, true>(s, blocks, cfg, sc, io, count);
    else launch_own<MODE_EXTEND, false>(s, blocks, cfg, sc, io, count);
}

void pt_launch_shadow_own(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                          DevShadow sh, const uint32_t *shadow_queue, const uint32_t *count, uint8_t *occ) {
    if (occ) {                                  // ptmi_debug_occluded (never with a queue)
        OccludedIO io{sh.SO, occ, sh.cap};
        if (cfg.cull) launch_own<MODE_SHADOW, true>(s, blocks, cfg, sc, io, count);
        else launch_own<MODE_SHADOW, false>(s, blocks, cfg, sc, io, count);
        return;
    }
    ShadowIO io{p.L, sh.SO, shadow_queue, p.l_stride, sh.cap};
    if (cfg.cull) launch_own<MODE_SHADOW, true>(s, blocks, cfg, sc, io, count);
    else launch_own<MODE_SHADOW, false>(s, blocks, cfg, sc, io, count);
}

#ifdef PT_UTIL_STATS
int pt_util_read_own(unsigned long long *h32, int reset) {
    if (hipMemcpyFromSymbol(h32, HIP_SYMBOL(g_util), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_util), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif
