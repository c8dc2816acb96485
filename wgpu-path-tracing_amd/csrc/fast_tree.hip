// fast_tree.hip — host code: a traversal hierarchy rebuilt over the REFERENCE's leaves.
//
// Why this is allowed: the reference tests a triangle iff the slab predicate of
// pt.wgsl:234-245 passes for every box on the path from the root to the triangle's leaf. With the
// contract's slab arithmetic, (bound - o) * (1/d) followed by min/max, the predicate is monotone under
// box containment for every "regular" ray (all three 1/d finite and non-zero): a larger box yields a
// per-axis interval that contains the smaller box's interval (IEEE subtraction and multiplication are
// monotone, no NaN can arise), so the smaller box passing implies the larger one passing. The
// reference's inner boxes are exact unions of their children (bvh.ts:14-28 recomputes min/max per
// range; checked at upload), hence
//     leaves the reference visits  =  { leaf : its OWN box passes }  for regular rays,
// and ANY hierarchy whose inner boxes are exact unions of the same leaf boxes visits the same leaves
// and returns the same (t, triangle, u, v). The reference builder only tries 11 equal-count splits
// on the longest axis; a full-sweep SAH tree over the same leaves halves the boxes a Cornell ray
// tests (25.7 -> 13.6 per ray, measured). Irregular rays (a zero, subnormal or non-finite direction
// component) keep walking the reference's own tree (traverse.hip).
#include "fast_tree.h"
#include "pt_device.h"
#include <limits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <future>
#include <numeric>
#include <thread>

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = INFINITY; mx[k] = -INFINITY; } }
    void grow(const float *lo, const float *hi) {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], lo[k]); mx[k] = std::max(mx[k], hi[k]); }
    }
    void grow(const Box &o) { grow(o.mn, o.mx); }
    double area() const {
        double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct Builder {
    const std::vector<PtFastLeaf> &leaves;
    std::vector<uint32_t> idx;            // permutation of leaf ids; a node owns a contiguous range
    std::vector<float> cen[3];
    std::vector<float4> &out;             // pre-sized: a tree over n leaves has n - 1 internal nodes
    std::atomic<uint32_t> depth{0};
    std::atomic<int> spare_threads{0};    // how many more subtree tasks may run beside their parent
    uint32_t depth_limit = 60;            // levels of internal nodes + leaves the tree may have (the traversal stacks' budget)
    uint32_t sweep_max = 4096;            // ranges up to this many leaves get the full sweep on every axis, larger ones 32 bins per axis

    Box range_box(uint32_t s, uint32_t e) const {
        Box b; b.reset();
        for (uint32_t i = s; i < e; i++) b.grow(leaves[idx[i]].mn, leaves[idx[i]].mx);
        return b;
    }

    // best split of [s, e): returns the split position after reordering idx[s..e) along the chosen axis
    uint32_t split(uint32_t s, uint32_t e, uint32_t d) {
        const uint32_t n = e - s;
        double best_cost = INFINITY; int best_axis = -1; uint32_t best_pos = s + n / 2;
        // keep the tree within the traversal stack: near the depth limit fall back to halving
        uint32_t lg = 0; while ((1u << lg) < n) lg++;
        if (d + lg >= depth_limit) return s + n / 2;
        if (n <= sweep_max) {
            // full sweep on every axis
            std::vector<uint32_t> order(n);
            std::vector<Box> suffix(n + 1);
            for (int ax = 0; ax < 3; ax++) {
                for (uint32_t i = 0; i < n; i++) order[i] = idx[s + i];
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cen[ax][a] < cen[ax][b]; });
                suffix[n].reset();
                for (uint32_t i = n; i-- > 0;) { suffix[i] = suffix[i + 1]; suffix[i].grow(leaves[order[i]].mn, leaves[order[i]].mx); }
                Box pre; pre.reset();
                uint64_t wl = 0, wt = 0;
                for (uint32_t i = 0; i < n; i++) wt += leaves[order[i]].weight;
                for (uint32_t i = 1; i < n; i++) {
                    pre.grow(leaves[order[i - 1]].mn, leaves[order[i - 1]].mx);
                    wl += leaves[order[i - 1]].weight;
                    double cost = pre.area() * (double)wl + suffix[i].area() * (double)(wt - wl);
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_pos = s + i; }
                }
            }
            if (best_axis >= 0)
                std::stable_sort(idx.begin() + s, idx.begin() + e,
                                 [&](uint32_t a, uint32_t b) { return cen[best_axis][a] < cen[best_axis][b]; });
            return best_pos;
        }
        // large ranges: 32 centroid bins per axis
        constexpr int NB = 32;
        Box cb; cb.reset();
        for (uint32_t i = s; i < e; i++) { float c3[3] = {cen[0][idx[i]], cen[1][idx[i]], cen[2][idx[i]]}; cb.grow(c3, c3); }
        float best_plane = 0.0f;
        for (int ax = 0; ax < 3; ax++) {
            float lo = cb.mn[ax], ext = cb.mx[ax] - cb.mn[ax];
            if (!(ext > 0.0f)) continue;
            Box bins[NB]; uint64_t bw[NB];
            for (int k = 0; k < NB; k++) { bins[k].reset(); bw[k] = 0; }
            for (uint32_t i = s; i < e; i++) {
                const PtFastLeaf &l = leaves[idx[i]];
                int k = std::min(NB - 1, (int)((cen[ax][idx[i]] - lo) / ext * NB));
                bins[k].grow(l.mn, l.mx); bw[k] += l.weight;
            }
            Box suf[NB + 1]; uint64_t sw[NB + 1];
            suf[NB].reset(); sw[NB] = 0;
            for (int k = NB; k-- > 0;) { suf[k] = suf[k + 1]; suf[k].grow(bins[k]); sw[k] = sw[k + 1] + bw[k]; }
            Box pre; pre.reset(); uint64_t pw = 0;
            for (int k = 1; k < NB; k++) {
                pre.grow(bins[k - 1]); pw += bw[k - 1];
                if (pw == 0 || sw[k] == 0) continue;
                double cost = pre.area() * (double)pw + suf[k].area() * (double)sw[k];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_plane = lo + ext * (float)k / NB; }
            }
        }
        if (best_axis < 0) {                // all centroids coincide: split in the middle
            return s + n / 2;
        }
        auto mid = std::stable_partition(idx.begin() + s, idx.begin() + e,
                                         [&](uint32_t a) { return cen[best_axis][a] < best_plane; });
        uint32_t pos = (uint32_t)(mid - idx.begin());
        if (pos == s || pos == e) pos = s + n / 2;
        return pos;
    }

    // returns the child reference of range [s, e) and writes its box. Internal nodes are numbered in preorder: the node of
    // [s, e) has index `me`, its left subtree (pos - s leaves, pos - s - 1 internal nodes) starts at me + 1, its right subtree
    // at me + (pos - s). Subtrees own disjoint parts of idx and of out, so large ones are built by other threads.
    uint32_t build(uint32_t s, uint32_t e, uint32_t d, uint32_t me, Box &box) {
        uint32_t seen = depth.load(std::memory_order_relaxed);
        while (seen < d && !depth.compare_exchange_weak(seen, d, std::memory_order_relaxed)) {}
        if (e - s == 1) {
            const PtFastLeaf &l = leaves[idx[s]];
            box.reset(); box.grow(l.mn, l.mx);
            return l.ref;
        }
        const uint32_t pos = split(s, e, d);
        Box lb, rb;
        uint32_t lref, rref;
        if (e - s >= 16384 && spare_threads.fetch_sub(1, std::memory_order_relaxed) > 0) {
            auto left = std::async(std::launch::async, [&] { return build(s, pos, d + 1, me + 1, lb); });
            rref = build(pos, e, d + 1, me + (pos - s), rb);
            lref = left.get();
            spare_threads.fetch_add(1, std::memory_order_relaxed);
        } else {
            if (e - s >= 16384) spare_threads.fetch_add(1, std::memory_order_relaxed);       // undo the claim that found none
            lref = build(s, pos, d + 1, me + 1, lb);
            rref = build(pos, e, d + 1, me + (pos - s), rb);
        }
        float fl, fr; std::memcpy(&fl, &lref, 4); std::memcpy(&fr, &rref, 4);
        float4 *w = &out[(size_t)me * 4];
        w[0] = make_float4(lb.mn[0], lb.mn[1], lb.mn[2], lb.mx[0]);
        w[1] = make_float4(lb.mx[1], lb.mx[2], rb.mn[0], rb.mn[1]);
        w[2] = make_float4(rb.mn[2], rb.mx[0], rb.mx[1], rb.mx[2]);
        w[3] = make_float4(fl, fr, 0.0f, 0.0f);
        box = lb; box.grow(rb);
        return me;
    }
};

}  // namespace

// ---- tree rotations -------------------------------------------------------------------------------------------------
// The greedy top-down SAH build is locally optimal per split, not globally. Afterwards every internal node N = (A, B) is
// offered the four classic rotations — swap B with a child of A, or A with a child of B — and takes the one that shrinks the
// surface area of the rebuilt child most (Kensler 2008); a few bottom-up passes. Any topology is as good as any other for
// the results (inner boxes stay exact unions of the same leaf boxes); a smaller summed area means fewer box
// tests per ray. The pass works on the wide nodes in place: a child reference moves, the preorder numbers stay (they are
// names, no kernel relies on their order), boxes are re-united bottom-up, and the depth is recomputed (a rotation may
// deepen a path by one level; the caller's limit is checked again).
namespace {

struct WideView;
uint32_t depth_of(const WideView &v, uint32_t n);

struct WideView {
    std::vector<float4> &w;
    static bool leaf(uint32_t ref) { return (ref & PT_REF_LEAF) != 0u; }
    uint32_t ref(uint32_t n, int side) const { uint32_t r; std::memcpy(&r, side ? &w[(size_t)n * 4 + 3].y : &w[(size_t)n * 4 + 3].x, 4); return r; }
    void set_ref(uint32_t n, int side, uint32_t r) { std::memcpy(side ? &w[(size_t)n * 4 + 3].y : &w[(size_t)n * 4 + 3].x, &r, 4); }
    Box box(uint32_t n, int side) const {
        const float4 *q = &w[(size_t)n * 4];
        Box b;
        if (side == 0) { b.mn[0] = q[0].x; b.mn[1] = q[0].y; b.mn[2] = q[0].z; b.mx[0] = q[0].w; b.mx[1] = q[1].x; b.mx[2] = q[1].y; }
        else { b.mn[0] = q[1].z; b.mn[1] = q[1].w; b.mn[2] = q[2].x; b.mx[0] = q[2].y; b.mx[1] = q[2].z; b.mx[2] = q[2].w; }
        return b;
    }
    void set_box(uint32_t n, int side, const Box &b) {
        float4 *q = &w[(size_t)n * 4];
        if (side == 0) { q[0] = make_float4(b.mn[0], b.mn[1], b.mn[2], b.mx[0]); q[1].x = b.mx[1]; q[1].y = b.mx[2]; }
        else { q[1].z = b.mn[0]; q[1].w = b.mn[1]; q[2] = make_float4(b.mn[2], b.mx[0], b.mx[1], b.mx[2]); }
    }
};

// one bottom-up pass over the subtree of n (at `level`, the root at 1); returns the subtree's height in levels (a leaf = 1) and
// adds the rotations made to `made`. A rotation is only taken if no leaf ends up deeper than max_depth.
uint32_t rotate_pass(WideView &v, uint32_t n, uint32_t level, uint32_t max_depth, size_t &made) {
    uint32_t h[2];
    for (int side = 0; side < 2; side++) {
        const uint32_t c = v.ref(n, side);
        h[side] = WideView::leaf(c) ? 1u : rotate_pass(v, c, level + 1, max_depth, made);
    }
    // children boxes may have changed below: refresh this node's copies of them
    for (int side = 0; side < 2; side++) {
        const uint32_t c = v.ref(n, side);
        if (!WideView::leaf(c)) { Box b = v.box(c, 0); b.grow(v.box(c, 1)); v.set_box(n, side, b); }
    }
    auto height = [&](uint32_t ref) -> uint32_t {            // height of a grandchild: one recursion level, cheap enough
        if (WideView::leaf(ref)) return 1u;
        uint32_t d = 0;
        for (int side = 0; side < 2; side++) { const uint32_t c = v.ref(ref, side); d = std::max(d, WideView::leaf(c) ? 1u : depth_of(v, c)); }
        return d + 1;
    };
    // candidate: child A = ref(n, a) internal with children (A0, A1), sibling B = ref(n, 1 - a):
    // swap B with A_k  ->  A' = (B, A_{1-k}), n' = (A', A_k). Only A's area changes.
    double best_gain = 0.0; int best_a = -1, best_k = -1; uint32_t best_h = 0;
    for (int a = 0; a < 2; a++) {
        const uint32_t A = v.ref(n, a);
        if (WideView::leaf(A)) continue;
        const Box boxA = v.box(n, a), boxB = v.box(n, 1 - a);
        for (int k = 0; k < 2; k++) {
            Box merged = boxB; merged.grow(v.box(A, 1 - k));
            const double gain = boxA.area() - merged.area();
            if (!(gain > best_gain && gain > 1e-9 * boxA.area())) continue;
            const uint32_t hB = h[1 - a], hK = height(v.ref(A, k)), hO = height(v.ref(A, 1 - k));
            const uint32_t new_h = 1u + std::max(1u + std::max(hB, hO), hK);          // n' = (A' = (B, A_other), A_k)
            if (level + new_h - 1u > max_depth) continue;
            best_gain = gain; best_a = a; best_k = k; best_h = new_h;
        }
    }
    if (best_a < 0) return 1u + std::max(h[0], h[1]);
    const int a = best_a, k = best_k;
    const uint32_t A = v.ref(n, a), B = v.ref(n, 1 - a), Ak = v.ref(A, k);
    const Box boxB = v.box(n, 1 - a), boxAk = v.box(A, k);
    v.set_ref(A, k, B); v.set_box(A, k, boxB);                   // B goes down into A
    v.set_ref(n, 1 - a, Ak); v.set_box(n, 1 - a, boxAk);         // A_k comes up beside A
    Box merged = v.box(A, 0); merged.grow(v.box(A, 1));
    v.set_box(n, a, merged);
    made++;
    return best_h;
}

uint32_t depth_of(const WideView &v, uint32_t n) {
    uint32_t d = 0;
    for (int side = 0; side < 2; side++) {
        const uint32_t c = v.ref(n, side);
        d = std::max(d, WideView::leaf(c) ? 1u : depth_of(v, c));
    }
    return d + 1;
}

double summed_area(const WideView &v, uint32_t n) {          // SAH proxy: sum of all child-box areas below n
    double s = 0.0;
    for (int side = 0; side < 2; side++) {
        s += v.box(n, side).area();
        const uint32_t c = v.ref(n, side);
        if (!WideView::leaf(c)) s += summed_area(v, c);
    }
    return s;
}

}  // namespace

#ifndef PT_TREE_ROTATIONS
#define PT_TREE_ROTATIONS 4          /* bottom-up passes; 0 = off */
#endif
#ifndef PT_TREE_ROTATION_DEPTH
#define PT_TREE_ROTATION_DEPTH 0     /* 0 = the rule below; otherwise a fixed depth budget (experiments) */
#endif

namespace {

// the greedy top-down build (no rotations): n - 1 wide nodes in preorder over n leaves
void build_hierarchy(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref, uint32_t &depth,
                     uint32_t depth_limit, uint32_t sweep_max = 4096) {
    wnodes.clear();
    Builder b{leaves, {}, {}, wnodes};
    b.depth_limit = depth_limit; b.sweep_max = sweep_max;
    const uint32_t n = (uint32_t)leaves.size();
    const unsigned hw = std::thread::hardware_concurrency();
    b.spare_threads = (int)std::min(15u, hw > 1 ? hw - 1 : 0u);
    b.idx.resize(n);
    std::iota(b.idx.begin(), b.idx.end(), 0u);
    for (int k = 0; k < 3; k++) {
        b.cen[k].resize(n);
        for (uint32_t i = 0; i < n; i++) b.cen[k][i] = 0.5f * leaves[i].mn[k] + 0.5f * leaves[i].mx[k];
    }
    wnodes.assign(n > 1 ? (size_t)(n - 1) * 4 : 0, make_float4(0, 0, 0, 0));
    Box root;
    root_ref = b.build(0, n, 1, 0, root);
    depth = b.depth.load();
}

// bottom-up rotation passes over a finished tree of n_leaves leaves (any leaf references); depth is updated
void rotate_tree(std::vector<float4> &wnodes, uint32_t root_ref, uint32_t n_leaves, uint32_t &depth, uint32_t depth_cap) {
    if (!(PT_TREE_ROTATIONS > 0 && n_leaves > 2 && n_leaves <= 65536u && !(root_ref & PT_REF_LEAF))) return;     // (334 174 leaves: -1 % area for +0.2 s)
    const std::vector<float4> before = wnodes;
    WideView v{wnodes};
    const bool dbg = std::getenv("PTMI_TREE_DEBUG") != nullptr;
    const double a0 = dbg ? summed_area(v, root_ref) : 0.0;
    // Depth budget: the LDS kernels hold a whole node stack per lane (trees of up to 14 levels run two workgroups per
    // CU), so a small tree may not grow past 14 levels; deeper ones (spilling stacks) get four more levels.
    const uint32_t max_depth = PT_TREE_ROTATION_DEPTH ? PT_TREE_ROTATION_DEPTH : (depth <= 14u ? 14u : std::min(depth_cap, depth + 4u));
    size_t total = 0;
    for (int pass = 0; pass < PT_TREE_ROTATIONS; pass++) {
        size_t made = 0;
        rotate_pass(v, root_ref, 1u, max_depth, made);
        total += made;
        if (made == 0) break;
    }
    const uint32_t d = depth_of(v, root_ref);
    if (dbg) std::fprintf(stderr, "fast tree: %u leaves, %zu rotations, summed box area %.6g -> %.6g (%.1f %%), depth %u -> %u\n",
                          n_leaves, total, a0, summed_area(v, root_ref), 100.0 * (summed_area(v, root_ref) / a0 - 1.0), depth, d);
    if (d > depth_cap) wnodes = before;                 // keep within the traversal's depth limit
    else depth = d;
}

}  // namespace

void pt_build_fast_tree(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref,
                        uint32_t &depth) {
    build_hierarchy(leaves, wnodes, root_ref, depth, 60u);
    rotate_tree(wnodes, root_ref, (uint32_t)leaves.size(), depth, 60u);
}

// ---- own leaves -----------------------------------------------------------------------------------------------------
#ifndef PT_OWN_C_BOX
#define PT_OWN_C_BOX 1.0             /* cost of a box-pair step (the unit) */
#endif
#ifndef PT_OWN_C_TRI
#define PT_OWN_C_TRI 0.9             /* ... of one triangle test (54 against ~60 vector instructions) */
#endif
#ifndef PT_OWN_C_OPEN
#define PT_OWN_C_OPEN 0.35           /* ... of opening a leaf (its entry leaves the lane's list, the loop is set up, a share of a vote) */
#endif
#ifndef PT_OWN_PAD_LOG2
#define PT_OWN_PAD_LOG2 (-16)        /* boxes grow by 2^this x the largest coordinate magnitude of the scene */
#endif

namespace {

double env_or(const char *name, double dflt) {          // experiments only (tools/own_leaf_gate.py)
    const char *e = std::getenv(name);
    return e ? std::atof(e) : dflt;
}

struct Collapse {
    const std::vector<float4> &w;          // the per-triangle hierarchy
    uint32_t max_leaf;
    double c_box, c_tri, c_open;
    std::vector<uint8_t> leaf;             // internal node -> becomes a leaf
    std::vector<uint32_t> count;           // triangles below an internal node
    struct Sub { uint32_t n; double cost; };
    Sub visit(uint32_t ref, uint32_t level) {
        if (ref & PT_REF_LEAF) return {1u, c_open + c_tri};
        WideView v{const_cast<std::vector<float4> &>(w)};
        const Box bl = v.box(ref, 0), br = v.box(ref, 1);
        Box all = bl; all.grow(br);
        const Sub l = visit(v.ref(ref, 0), level + 1), r = visit(v.ref(ref, 1), level + 1);
        const double a = all.area();
        const double inner = c_box + (a > 0.0 ? (bl.area() * l.cost + br.area() * r.cost) / a : l.cost + r.cost);
        const uint32_t n = l.n + r.n;
        const double as_leaf = c_open + c_tri * n;
        count[ref] = n;
        if (n <= max_leaf && as_leaf <= inner) { leaf[ref] = 1; return {n, as_leaf}; }
        return {n, inner};
    }
};

}  // namespace

bool pt_build_own_tree(const ptmi_triangle *tris, const std::vector<uint32_t> &which, uint32_t max_leaf, uint32_t depth_limit,
                       PtOwnTree &out) {
    out = PtOwnTree();
    const uint32_t n = (uint32_t)which.size();
    if (n == 0) return false;
    max_leaf = std::max(1u, std::min(max_leaf, PT_LEAF_MAX_TRIS));
    // one "leaf" per triangle: the box of its three vertices and of v0 + e1, v0 + e2 as the intersection test sees them
    std::vector<PtFastLeaf> units(n);
    double biggest = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        const ptmi_triangle &t = tris[which[i]];
        PtFastLeaf &u = units[i];
        for (int k = 0; k < 3; k++) {
            const float a = t.v0[k], b = t.v1[k], c = t.v2[k];
            const float b2 = a + (b - a), c2 = a + (c - a);
            if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(c) || !std::isfinite(b2) || !std::isfinite(c2)) return false;
            u.mn[k] = std::min(std::min(std::min(a, b), std::min(c, b2)), c2);
            u.mx[k] = std::max(std::max(std::max(a, b), std::max(c, b2)), c2);
            biggest = std::max(biggest, std::max(std::fabs((double)u.mn[k]), std::fabs((double)u.mx[k])));
        }
        u.ref = PT_REF_LEAF | i; u.weight = 1u;
    }
    const bool dbg = std::getenv("PTMI_TREE_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    std::vector<float4> per_tri;
    uint32_t root = PT_REF_NONE, depth = 0;
    // (scenes of a few thousand triangles get the full sweep throughout; large ones bins down to ranges of 128: the 1 M-triangle scene
    // builds in a third of the time for +1 % box steps)
    build_hierarchy(units, per_tri, root, depth, depth_limit, n <= 8192u ? 4096u : (uint32_t)env_or("PTMI_OWN_SWEEP_MAX", 128));
    const auto t1 = now();
    // collapse subtrees into leaves where that is cheaper
    Collapse col{per_tri, max_leaf, env_or("PTMI_OWN_C_BOX", PT_OWN_C_BOX), env_or("PTMI_OWN_C_TRI", PT_OWN_C_TRI),
                 env_or("PTMI_OWN_C_OPEN", PT_OWN_C_OPEN), {}, {}};
    col.leaf.assign(n > 1 ? n - 1 : 0, 0); col.count.assign(n > 1 ? n - 1 : 0, 0);
    col.visit(root, 1);
    const auto t2 = now();
    // emit: surviving internal nodes in preorder, the triangles in the order their leaves hang off them
    out.tripos.reserve((size_t)n * 3);
    WideView v{per_tri};
    auto put_tri = [&](uint32_t unit) {
        const uint32_t orig = which[unit];
        const ptmi_triangle &t = tris[orig];
        float w; std::memcpy(&w, &orig, 4);
        out.tripos.push_back(make_float4(t.v0[0], t.v0[1], t.v0[2], w));
        out.tripos.push_back(make_float4(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], 0.0f));
        out.tripos.push_back(make_float4(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], 0.0f));
    };
    std::vector<uint32_t> todo;                     // triangles of a subtree, left first
    auto leaf_of = [&](uint32_t ref) -> uint32_t {
        const uint32_t first = (uint32_t)(out.tripos.size() / 3);
        uint32_t cnt = 0;
        todo.clear(); todo.push_back(ref);
        while (!todo.empty()) {
            const uint32_t r = todo.back(); todo.pop_back();
            if (r & PT_REF_LEAF) { put_tri(r & PT_LEAF_OFF_MASK); cnt++; }
            else { todo.push_back(v.ref(r, 1)); todo.push_back(v.ref(r, 0)); }
        }
        out.n_leaves++; out.max_leaf_tris = std::max(out.max_leaf_tris, cnt);
        return PT_REF_LEAF | ((cnt - 1u) << PT_LEAF_OFF_BITS) | first;
    };
    auto is_leaf = [&](uint32_t ref) { return (ref & PT_REF_LEAF) || col.leaf[ref]; };
    if (is_leaf(root)) {
        out.root_ref = leaf_of(root);
        out.depth = 1;
    } else {
        // preorder numbers first (a node's number is known before its subtree is emitted), then one pass that fills them
        struct Item { uint32_t old_node, new_node; };
        std::vector<uint32_t> new_of(per_tri.size() / 4, PT_REF_NONE);
        {
            std::vector<uint32_t> st{root};
            uint32_t next = 0;
            while (!st.empty()) {
                const uint32_t r = st.back(); st.pop_back();
                new_of[r] = next++;
                const uint32_t a = v.ref(r, 0), b = v.ref(r, 1);
                if (!is_leaf(b)) st.push_back(b);
                if (!is_leaf(a)) st.push_back(a);
            }
            out.wnodes.assign((size_t)next * 4, make_float4(0, 0, 0, 0));
        }
        std::vector<uint32_t> st{root};
        while (!st.empty()) {                        // the same preorder: leaves are numbered in the order a left-first walk meets them
            const uint32_t r = st.back(); st.pop_back();
            float4 *w = &out.wnodes[(size_t)new_of[r] * 4];
            const float4 *q = &per_tri[(size_t)r * 4];
            w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
            uint32_t refs[2];
            for (int side = 0; side < 2; side++) {
                const uint32_t c = v.ref(r, side);
                refs[side] = is_leaf(c) ? leaf_of(c) : new_of[c];
            }
            float fl, fr; std::memcpy(&fl, &refs[0], 4); std::memcpy(&fr, &refs[1], 4);
            w[3] = make_float4(fl, fr, 0.0f, 0.0f);
            const uint32_t a = v.ref(r, 0), b = v.ref(r, 1);
            if (!is_leaf(b)) st.push_back(b);
            if (!is_leaf(a)) st.push_back(a);
        }
        out.root_ref = 0u;
        WideView nv{out.wnodes};
        out.depth = depth_of(nv, 0u);
        rotate_tree(out.wnodes, out.root_ref, out.n_leaves, out.depth, depth_limit);
    }
    const auto t3 = now();
    // padding (see the header): every child box, and the root box the kernels test first
    const float pad = std::max((float)std::ldexp(biggest, PT_OWN_PAD_LOG2 + (int)env_or("PTMI_OWN_PAD_EXTRA_LOG2", 0)), std::numeric_limits<float>::min());
    if (!std::isfinite(pad)) return false;
    auto lower = [&](float x) { const float y = x - pad; return y < x ? y : std::nextafterf(x, -INFINITY); };
    auto upper = [&](float x) { const float y = x + pad; return y > x ? y : std::nextafterf(x, INFINITY); };
    Box rb; rb.reset();
    if (out.root_ref & PT_REF_LEAF) { for (const PtFastLeaf &u : units) rb.grow(u.mn, u.mx); }
    else { WideView nv{out.wnodes}; rb = nv.box(0, 0); rb.grow(nv.box(0, 1)); }
    for (int k = 0; k < 3; k++) { out.root_min[k] = lower(rb.mn[k]); out.root_max[k] = upper(rb.mx[k]); }
    {
        WideView nv{out.wnodes};
        for (uint32_t i = 0; i < out.wnodes.size() / 4; i++)
            for (int side = 0; side < 2; side++) {
                Box b = nv.box(i, side);
                for (int k = 0; k < 3; k++) { b.mn[k] = lower(b.mn[k]); b.mx[k] = upper(b.mx[k]); }
                nv.set_box(i, side, b);
            }
    }
    for (int k = 0; k < 3; k++) if (!std::isfinite(out.root_min[k]) || !std::isfinite(out.root_max[k])) return false;
    if (dbg) std::fprintf(stderr, "own tree: %u triangles -> %zu nodes, %u leaves, depth %u; hierarchy %.1f ms, collapse %.1f, emit + rotations %.1f, padding %.1f\n",
                          n, out.wnodes.size() / 4, out.n_leaves, out.depth, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    out.pad = pad;
    out.safe_origin = (float)std::min(8.0 * biggest, 3.0e38);      // the rounding of the fused tests stays below a quarter of the padding up to here
    return true;
}

// ---- quantised image ------------------------------------------------------------------------------------------------
bool pt_quantize_tree(const std::vector<PtFastLeaf> &leaves, const std::vector<float4> &wnodes, const std::vector<float4> &tripos,
                      std::vector<uint4> &qnodes, std::vector<uint32_t> &stream, float origin[3], float scale[3],
                      uint32_t top_nodes, uint32_t &n_top) {
    qnodes.clear(); stream.clear(); n_top = 0;
    const size_t n_nodes = wnodes.size() / 4;
    if (n_nodes == 0 || leaves.empty()) return false;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t n_tri_refs = 0;
    for (const PtFastLeaf &l : leaves) {
        for (int k = 0; k < 3; k++) {
            if (!std::isfinite(l.mn[k]) || !std::isfinite(l.mx[k]) || l.mn[k] > l.mx[k]) return false;
            mn[k] = std::min(mn[k], l.mn[k]); mx[k] = std::max(mx[k], l.mx[k]);
        }
        n_tri_refs += l.weight;
    }
    if (leaves.size() * 8 + n_tri_refs * 9 >= (1ull << 31)) return false;
    for (int k = 0; k < 3; k++) {
        origin[k] = mn[k];
        const double ext = (double)mx[k] - (double)mn[k];
        float s = (float)(ext / 65535.0);
        if (!std::isfinite(s)) return false;
        if (ext > 0.0) {
            if (!(s > 0.0f)) s = std::numeric_limits<float>::denorm_min();
            int guard = 0;
            while (std::fmaf(s, 65535.0f, origin[k]) < mx[k] && guard++ < 64) s = std::nextafterf(s, INFINITY);   // the last plane reaches the far side
            if (std::fmaf(s, 65535.0f, origin[k]) < mx[k]) return false;
        }
        scale[k] = s;
    }
    auto plane_lo = [&](int k, float v) -> uint32_t {         // largest plane number whose plane is <= v
        if (!(scale[k] > 0.0f)) return 0u;
        double q = std::floor(((double)v - (double)origin[k]) / (double)scale[k]);
        uint32_t u = q <= 0.0 ? 0u : q >= 65535.0 ? 65535u : (uint32_t)q;
        while (u > 0u && std::fmaf(scale[k], (float)u, origin[k]) > v) u--;
        return u;
    };
    auto plane_hi = [&](int k, float v) -> uint32_t {         // smallest plane number whose plane is >= v
        if (!(scale[k] > 0.0f)) return 0u;
        double q = std::ceil(((double)v - (double)origin[k]) / (double)scale[k]);
        uint32_t u = q <= 0.0 ? 0u : q >= 65535.0 ? 65535u : (uint32_t)q;
        while (u < 65535u && std::fmaf(scale[k], (float)u, origin[k]) < v) u++;
        return u;
    };
    // the leaf stream, in the order the leaves hang off the preorder nodes (neighbours in the tree are neighbours in memory).
    // Pass 1 assigns every leaf child its place (sequential, two words per node); pass 2 fills nodes and stream in parallel.
    const size_t n_tris = tripos.size() / 3;
    std::vector<uint32_t> leaf_of_first(n_tris, 0xFFFFFFFFu);         // a leaf is identified by its first triangle
    for (size_t i = 0; i < leaves.size(); i++) {
        const uint32_t first = leaves[i].ref & PT_LEAF_OFF_MASK;
        if (first >= n_tris) return false;
        leaf_of_first[first] = (uint32_t)i;
    }
    std::vector<uint32_t> child_off(n_nodes * 2, 0u);
    size_t total = 0;
    for (size_t i = 0; i < n_nodes; i++) {
        uint32_t refs[2]; std::memcpy(&refs[0], &wnodes[i * 4 + 3].x, 4); std::memcpy(&refs[1], &wnodes[i * 4 + 3].y, 4);
        for (int c = 0; c < 2; c++)
            if (refs[c] & PT_REF_LEAF) {
                child_off[i * 2 + c] = (uint32_t)total;
                total += 8 + 9 * (size_t)(((refs[c] >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u);
            }
    }
    if (total >= (1ull << 31)) return false;
    stream.assign(total, 0u);
    qnodes.resize(n_nodes * 2);
    // new numbers: the top of the tree breadth-first (root first), then everything else in preorder
    std::vector<uint32_t> renum(n_nodes, 0xFFFFFFFFu);
    {
        std::vector<uint32_t> bfs; bfs.reserve(top_nodes);
        bfs.push_back(0u);
        for (size_t h = 0; h < bfs.size() && bfs.size() < top_nodes; h++) {
            uint32_t refs[2]; std::memcpy(&refs[0], &wnodes[(size_t)bfs[h] * 4 + 3].x, 4); std::memcpy(&refs[1], &wnodes[(size_t)bfs[h] * 4 + 3].y, 4);
            for (int c = 0; c < 2 && bfs.size() < top_nodes; c++)
                if (!(refs[c] & PT_REF_LEAF)) bfs.push_back(refs[c]);
        }
        for (size_t k = 0; k < bfs.size(); k++) renum[bfs[k]] = (uint32_t)k;
        n_top = (uint32_t)bfs.size();
        uint32_t next = n_top;
        for (size_t i = 0; i < n_nodes; i++) if (renum[i] == 0xFFFFFFFFu) renum[i] = next++;
    }
    auto area = [](const float *l, const float *h) {
        const double x = (double)h[0] - l[0], y = (double)h[1] - l[1], z = (double)h[2] - l[2];
        return 2.0 * (x * y + y * z + z * x);
    };
    auto fill = [&](size_t i0, size_t i1, double &growth, size_t &grown, bool &ok) {
        for (size_t i = i0; i < i1; i++) {
            const float4 *w = &wnodes[i * 4];
            const float lo[2][3] = {{w[0].x, w[0].y, w[0].z}, {w[1].z, w[1].w, w[2].x}};
            const float hi[2][3] = {{w[0].w, w[1].x, w[1].y}, {w[2].y, w[2].z, w[2].w}};
            uint32_t refs[2]; std::memcpy(&refs[0], &w[3].x, 4); std::memcpy(&refs[1], &w[3].y, 4);
            for (int c = 0; c < 2; c++) {
                uint32_t ql[3], qh[3];
                for (int k = 0; k < 3; k++) { ql[k] = plane_lo(k, lo[c][k]); qh[k] = plane_hi(k, hi[c][k]); }
                uint32_t ref = refs[c];
                if (ref & PT_REF_LEAF) {
                    const uint32_t first = ref & PT_LEAF_OFF_MASK, cnt = ((ref >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u;
                    const uint32_t li = first < n_tris ? leaf_of_first[first] : 0xFFFFFFFFu;
                    if (li == 0xFFFFFFFFu || leaves[li].ref != ref || (size_t)first + cnt > n_tris) { ok = false; continue; }
                    const PtFastLeaf &l = leaves[li];
                    uint32_t *h = &stream[child_off[i * 2 + c]];
                    std::memcpy(h, l.mn, 12); h[3] = first; std::memcpy(h + 4, l.mx, 12); h[7] = cnt;
                    for (uint32_t t = 0; t < cnt; t++)
                        for (int j = 0; j < 3; j++) std::memcpy(h + 8 + 9 * t + 3 * j, &tripos[3 * (size_t)(first + t) + j], 12);
                    ref = PT_REF_LEAF | child_off[i * 2 + c];
                } else {
                    ref = renum[ref];
                }
                qnodes[(size_t)renum[i] * 2 + c] = make_uint4(ql[0] | (ql[1] << 16), ql[2] | (qh[0] << 16), qh[1] | (qh[2] << 16), ref);
                float dl[3], dh[3];
                for (int k = 0; k < 3; k++) { dl[k] = std::fmaf(scale[k], (float)ql[k], origin[k]); dh[k] = std::fmaf(scale[k], (float)qh[k], origin[k]); }
                const double a0 = area(lo[c], hi[c]);
                if (a0 > 0.0) { growth += std::min(area(dl, dh) / a0 - 1.0, 1e6); grown++; }
            }
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_thr = n_nodes < 65536 ? 1 : std::min<size_t>(16, hw ? hw : 1);
    std::vector<double> g(n_thr, 0.0); std::vector<size_t> gn(n_thr, 0); std::vector<char> oks(n_thr, 1);
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_thr; t++)
            pool.emplace_back([&, t] { bool ok = true; fill(n_nodes * t / n_thr, n_nodes * (t + 1) / n_thr, g[t], gn[t], ok); oks[t] = ok; });
        bool ok = true; fill(0, n_nodes / n_thr, g[0], gn[0], ok); oks[0] = ok;
        for (auto &th : pool) th.join();
    }
    double growth = 0.0; size_t grown = 0;       // mean relative growth of the child boxes' surface area
    for (size_t t = 0; t < n_thr; t++) { growth += g[t]; grown += gn[t]; if (!oks[t]) { qnodes.clear(); stream.clear(); return false; } }
    // One grid for the whole scene suits scenes whose boxes are not many orders of magnitude smaller than the scene.
    // Where they are (a chain of boxes shrinking geometrically), the rounded boxes would admit far more rays than the exact
    // ones: same results, much more work. Such scenes keep the exact image.
    if (grown && growth / (double)grown > 0.25) { qnodes.clear(); stream.clear(); return false; }
    return true;
}

bool pt_quantize_nodes(const std::vector<float4> &wnodes, std::vector<uint4> &qnodes, float origin[3], float scale[3],
                       uint32_t top_nodes, uint32_t &n_top) {
    qnodes.clear(); n_top = 0;
    const size_t n_nodes = wnodes.size() / 4;
    if (n_nodes == 0) return false;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n_nodes; i++) {
        const float4 *w = &wnodes[i * 4];
        const float lo[2][3] = {{w[0].x, w[0].y, w[0].z}, {w[1].z, w[1].w, w[2].x}};
        const float hi[2][3] = {{w[0].w, w[1].x, w[1].y}, {w[2].y, w[2].z, w[2].w}};
        for (int c = 0; c < 2; c++)
            for (int k = 0; k < 3; k++) {
                if (!std::isfinite(lo[c][k]) || !std::isfinite(hi[c][k]) || lo[c][k] > hi[c][k]) return false;
                mn[k] = std::min(mn[k], lo[c][k]); mx[k] = std::max(mx[k], hi[c][k]);
            }
    }
    for (int k = 0; k < 3; k++) {
        origin[k] = mn[k];
        const double ext = (double)mx[k] - (double)mn[k];
        float s = (float)(ext / 65535.0);
        if (!std::isfinite(s)) return false;
        if (ext > 0.0) {
            if (!(s > 0.0f)) s = std::numeric_limits<float>::denorm_min();
            int guard = 0;
            while (std::fmaf(s, 65535.0f, origin[k]) < mx[k] && guard++ < 64) s = std::nextafterf(s, INFINITY);
            if (std::fmaf(s, 65535.0f, origin[k]) < mx[k]) return false;
        }
        scale[k] = s;
    }
    auto plane_lo = [&](int k, float v) -> uint32_t {
        if (!(scale[k] > 0.0f)) return 0u;
        double q = std::floor(((double)v - (double)origin[k]) / (double)scale[k]);
        uint32_t u = q <= 0.0 ? 0u : q >= 65535.0 ? 65535u : (uint32_t)q;
        while (u > 0u && std::fmaf(scale[k], (float)u, origin[k]) > v) u--;
        return u;
    };
    auto plane_hi = [&](int k, float v) -> uint32_t {
        if (!(scale[k] > 0.0f)) return 0u;
        double q = std::ceil(((double)v - (double)origin[k]) / (double)scale[k]);
        uint32_t u = q <= 0.0 ? 0u : q >= 65535.0 ? 65535u : (uint32_t)q;
        while (u < 65535u && std::fmaf(scale[k], (float)u, origin[k]) < v) u++;
        return u;
    };
    std::vector<uint32_t> renum(n_nodes, 0xFFFFFFFFu);
    {
        std::vector<uint32_t> bfs; bfs.reserve(top_nodes);
        bfs.push_back(0u);
        for (size_t h = 0; h < bfs.size() && bfs.size() < top_nodes; h++) {
            uint32_t refs[2]; std::memcpy(&refs[0], &wnodes[(size_t)bfs[h] * 4 + 3].x, 4); std::memcpy(&refs[1], &wnodes[(size_t)bfs[h] * 4 + 3].y, 4);
            for (int c = 0; c < 2 && bfs.size() < top_nodes; c++)
                if (!(refs[c] & PT_REF_LEAF)) bfs.push_back(refs[c]);
        }
        for (size_t k = 0; k < bfs.size(); k++) renum[bfs[k]] = (uint32_t)k;
        n_top = (uint32_t)bfs.size();
        uint32_t next = n_top;
        for (size_t i = 0; i < n_nodes; i++) if (renum[i] == 0xFFFFFFFFu) renum[i] = next++;
    }
    qnodes.resize(n_nodes * 2);
    auto fill = [&](size_t i0, size_t i1, double &growth, size_t &grown) {
        for (size_t i = i0; i < i1; i++) {
            const float4 *w = &wnodes[i * 4];
            const float lo[2][3] = {{w[0].x, w[0].y, w[0].z}, {w[1].z, w[1].w, w[2].x}};
            const float hi[2][3] = {{w[0].w, w[1].x, w[1].y}, {w[2].y, w[2].z, w[2].w}};
            uint32_t refs[2]; std::memcpy(&refs[0], &w[3].x, 4); std::memcpy(&refs[1], &w[3].y, 4);
            for (int c = 0; c < 2; c++) {
                uint32_t ql[3], qh[3];
                float dl[3], dh[3];
                for (int k = 0; k < 3; k++) {
                    ql[k] = plane_lo(k, lo[c][k]); qh[k] = plane_hi(k, hi[c][k]);
                    dl[k] = std::fmaf(scale[k], (float)ql[k], origin[k]); dh[k] = std::fmaf(scale[k], (float)qh[k], origin[k]);
                }
                const uint32_t ref = (refs[c] & PT_REF_LEAF) ? refs[c] : renum[refs[c]];
                qnodes[(size_t)renum[i] * 2 + c] = make_uint4(ql[0] | (ql[1] << 16), ql[2] | (qh[0] << 16), qh[1] | (qh[2] << 16), ref);
                const double ax = (double)hi[c][0] - lo[c][0], ay = (double)hi[c][1] - lo[c][1], az = (double)hi[c][2] - lo[c][2];
                const double a0 = 2.0 * (ax * ay + ay * az + az * ax);
                const double bx = (double)dh[0] - dl[0], by = (double)dh[1] - dl[1], bz = (double)dh[2] - dl[2];
                if (a0 > 0.0) { growth += std::min(2.0 * (bx * by + by * bz + bz * bx) / a0 - 1.0, 1e6); grown++; }
            }
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_thr = n_nodes < 65536 ? 1 : std::min<size_t>(16, hw ? hw : 1);
    std::vector<double> g(n_thr, 0.0); std::vector<size_t> gn(n_thr, 0);
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_thr; t++) pool.emplace_back([&, t] { fill(n_nodes * t / n_thr, n_nodes * (t + 1) / n_thr, g[t], gn[t]); });
        fill(0, n_nodes / n_thr, g[0], gn[0]);
        for (auto &th : pool) th.join();
    }
    double growth = 0.0; size_t grown = 0;
    for (size_t t = 0; t < n_thr; t++) { growth += g[t]; grown += gn[t]; }
    if (grown && growth / (double)grown > 0.25) { qnodes.clear(); return false; }       // a 16-bit grid is too coarse for this scene's boxes
    return true;
}
