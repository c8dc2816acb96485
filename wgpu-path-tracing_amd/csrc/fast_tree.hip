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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

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
    std::vector<float4> &out;
    uint32_t depth = 0;

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
        if (d + lg >= 60) return s + n / 2;
        if (n <= 4096) {
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

    // returns the child reference of range [s, e) and writes its box; internal nodes are appended in preorder
    uint32_t build(uint32_t s, uint32_t e, uint32_t d, Box &box) {
        depth = std::max(depth, d);
        if (e - s == 1) {
            const PtFastLeaf &l = leaves[idx[s]];
            box.reset(); box.grow(l.mn, l.mx);
            return l.ref;
        }
        const uint32_t me = (uint32_t)(out.size() / 4);
        out.resize(out.size() + 4);
        const uint32_t pos = split(s, e, d);
        Box lb, rb;
        const uint32_t lref = build(s, pos, d + 1, lb);
        const uint32_t rref = build(pos, e, d + 1, rb);
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

void pt_build_fast_tree(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref,
                        uint32_t &depth) {
    wnodes.clear();
    Builder b{leaves, {}, {}, wnodes};
    const uint32_t n = (uint32_t)leaves.size();
    b.idx.resize(n);
    std::iota(b.idx.begin(), b.idx.end(), 0u);
    for (int k = 0; k < 3; k++) {
        b.cen[k].resize(n);
        for (uint32_t i = 0; i < n; i++) b.cen[k][i] = 0.5f * leaves[i].mn[k] + 0.5f * leaves[i].mx[k];
    }
    wnodes.reserve((size_t)n * 4);
    Box root;
    root_ref = b.build(0, n, 1, root);
    depth = b.depth;
}
