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
#include "traverse_common.h"
#include <atomic>
#include <type_traits>

namespace {

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
template <int MODE, bool CULL, int STACK, bool SPILL, int REFILL, class Mem, class IO>
PT_DEV void trace_wave(const Mem &m, const DevScene &sc, const IO &io, uint32_t count, uint32_t gw,
                             uint32_t total_waves, uint32_t *stk, int stride, uint32_t *spill = nullptr,
                             uint32_t spill_lanes = 0) {
    constexpr bool ANY = MODE == MODE_SHADOW;
    constexpr int NODE_KEEP = ANY ? 2 : 3;
    const uint32_t lane = threadIdx.x & 63u;
    // gw (and so end, next) is the same in all 64 lanes; readfirstlane tells the compiler, which then keeps
    // the queue bookkeeping in SGPRs and turns the refill / exit tests into scalar branches
    gw = uniform(gw);
    const uint32_t ngroups = (count + 63u) >> 6;
    const uint32_t end = gw < ngroups ? ((ngroups - gw + total_waves - 1u) / total_waves) * 64u : 0u;
    uint32_t next = 0u;
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
        if (next < end && popc(act) <= REFILL) {
            const uint64_t idle = ~act;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t vi = next + rank;
            const uint32_t vslot = ((vi >> 6) * total_waves + gw) * 64u + (vi & 63u);
            if (!active && vi < end && vslot < count) {
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
            next += (uint32_t)__popcll(idle);
            UTIL(2, 1); UTIL(3, popc(ballot(active)) - popc(act));
            act = ballot(active);
        }
        if (act == 0ull && next >= end) break;
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
                                                         uint32_t *__restrict__ spill) {
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
        trace_wave<MODE, CULL, STACK, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK);
    } else {
        if (gw * 64u >= count) return;
        GlobalMem m{(glb_f4p)sc.wnodes, (glb_f4p)sc.tripos};
        trace_wave<MODE, CULL, STACK, true, PT_REFILL_GLOBAL>(m, sc, io, count, gw, gridDim.x * (GBLOCK / 64), stk + threadIdx.x, GBLOCK, sp, gridDim.x * GBLOCK);
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
template <int MODE, bool CULL, int STACK, bool TRIS_IN_LDS, bool SPILL, class IO>
__global__ __launch_bounds__(LBLOCK) PT_LDS_ATTR void k_trace_lds(DevScene sc, IO io, const uint32_t *__restrict__ count_ptr,
                                                      uint32_t *__restrict__ spill) {
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
    trace_wave<MODE, CULL, STACK, SPILL, REFILL_AT>(m, sc, io, count, gw, gridDim.x * (LBLOCK / 64), stk, LBLOCK,
                                           SPILL ? spill + (size_t)blockIdx.x * LBLOCK + threadIdx.x : nullptr, gridDim.x * LBLOCK);
}

template <int MODE, bool CULL, int STACK, bool TRIS, bool SPILL = false, class IO>
void launch_lds(hipStream_t s, int wgs, size_t bytes, const DevScene &sc, const IO &io, const uint32_t *count,
                uint32_t *spill = nullptr) {
    // the default dynamic-LDS cap is 64 KB; raise it once per instantiation and device
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_lds<MODE, CULL, STACK, TRIS, SPILL, IO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_trace_lds<MODE, CULL, STACK, TRIS, SPILL, IO>), dim3(wgs), dim3(LBLOCK), bytes, s, sc, io, count, spill);
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
    hipLaunchKernelGGL((k_trace_global<MODE, CULL, 16, QUANT, IO>), dim3(per_cu * cus), dim3(GBLOCK), 0, s, sc, io, count, spill);
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
        if (cfg.stack_entries <= 16) launch_lds<MODE, CULL, 16, true>(s, cus, bytes, sc, io, count);
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

size_t pt_spill_bytes(int blocks) {
    const int cus = blocks / 8 > 0 ? blocks / 8 : 1;
    return (size_t)GLOBAL_WGS_MAX * cus * GBLOCK * PT_SPILL_ENTRIES * sizeof(uint32_t);
}

#ifdef PT_UTIL_STATS
int pt_util_read_own(unsigned long long *h32, int reset);       // traverse_own.hip's counters
extern "C" __attribute__((visibility("default"))) int ptmi_debug_util_stats(unsigned long long *out32, int reset) {
    unsigned long long h[32], g[32];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_util), sizeof(h)) != hipSuccess) return 1;
    if (pt_util_read_own(g, reset)) return 1;
    for (int i = 0; i < 32; i++) out32[i] = h[i] + g[i];
    if (reset) { for (auto &x : h) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_util), h, sizeof(h)) != hipSuccess) return 1; }
    return 0;
}
#endif
