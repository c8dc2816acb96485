// scene_prep.cpp — host-side scene preparation behind include/ptmi_scene.h.
//
// Restates, in C++, the three producers of the hot path's input order:
//   * the partial quicksort of src/utils/arr.ts (not stable; its exact swap
//     sequence decides the triangle order inside equal-key runs),
//   * the SAH builder of src/renderer/bvh.ts (+ src/utils/aabb.ts),
//   * the emissive-light list of src/renderer/gpu.ts:121-138.
// JS numbers are doubles: centroids, extents and SAH costs are computed in double
// from f32 vertex data, AABB corners are stored as f32 (Float32Array).
#include "ptmi_scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

// ---------------------------------------------------------------- arr.ts ----
// Elements are moved by whole-value swaps/assignments exactly where arr.ts moves
// them; Cmp returns the sign-carrying difference like a JS comparator.
template <class T, class Cmp>
struct PartialSorter {
    T *a;
    Cmp cmp;

    void insertion(int64_t lo, int64_t hi) {                       // arr.ts:13-23
        for (int64_t i = lo + 1; i <= hi; i++) {
            T key = a[i];
            int64_t j = i - 1;
            while (j >= lo && cmp(a[j], key) > 0) { a[j + 1] = a[j]; j--; }
            a[j + 1] = key;
        }
    }
    int64_t median3(int64_t lo, int64_t hi) {                      // arr.ts:26-39
        int64_t mid = lo + ((hi - lo) >> 1);
        if (cmp(a[lo], a[mid]) > 0) std::swap(a[lo], a[mid]);
        if (cmp(a[mid], a[hi]) > 0) {
            std::swap(a[mid], a[hi]);
            if (cmp(a[lo], a[mid]) > 0) std::swap(a[lo], a[mid]);
        }
        return mid;
    }
    int64_t partition(int64_t lo, int64_t hi) {                    // arr.ts:41-65
        if (hi - lo > 10) {
            int64_t p = median3(lo, hi);
            std::swap(a[p], a[hi]);
        }
        T pivot = a[hi];
        int64_t i = lo - 1;
        for (int64_t j = lo; j < hi; j++) {
            if (cmp(a[j], pivot) <= 0) {
                i++;
                if (i != j) std::swap(a[i], a[j]);
            }
        }
        if (i + 1 != hi) std::swap(a[i + 1], a[hi]);
        return i + 1;
    }
    void run(int64_t start, int64_t end) {                         // arr.ts:68-108
        std::vector<int64_t> st;
        st.push_back(start); st.push_back(end - 1);
        while (!st.empty()) {
            int64_t hi = st.back(); st.pop_back();
            int64_t lo = st.back(); st.pop_back();
            if (hi - lo < 10) { insertion(lo, hi); continue; }
            if (!(lo < hi)) continue;
            int64_t p = partition(lo, hi);
            bool left_smaller = (p - lo) < (hi - p);
            if (left_smaller) {
                if (p + 1 < hi) { st.push_back(p + 1); st.push_back(hi); }
                if (p - 1 > lo) { st.push_back(lo); st.push_back(p - 1); }
            } else {
                if (p - 1 > lo) { st.push_back(lo); st.push_back(p - 1); }
                if (p + 1 < hi) { st.push_back(p + 1); st.push_back(hi); }
            }
        }
    }
};

struct CmpAsc { double operator()(double x, double y) const { return x - y; } };
struct CmpDesc { double operator()(double x, double y) const { return y - x; } };

// --------------------------------------------------------------- aabb.ts ----
struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; k++) {
            mn[k] = std::numeric_limits<float>::infinity();
            mx[k] = -std::numeric_limits<float>::infinity();
        }
    }
    // vec3.min / vec3.max component-wise (Math.min / Math.max: NaN poisons)
    static float jsmin(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
    static float jsmax(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
    void grow(const float *p) {
        for (int k = 0; k < 3; k++) { mn[k] = jsmin(p[k], mn[k]); mx[k] = jsmax(p[k], mx[k]); }
    }
    void grow(const ptmi_triangle &t) { grow(t.v0); grow(t.v1); grow(t.v2); }          // bvh.ts:18-25
    void merge(const Box &o) {
        for (int k = 0; k < 3; k++) { mn[k] = jsmin(o.mn[k], mn[k]); mx[k] = jsmax(o.mx[k], mx[k]); }
    }
    double area() const {                                                               // aabb.ts:43-48
        double dx = (double)mx[0] - (double)mn[0];
        double dy = (double)mx[1] - (double)mn[1];
        double dz = (double)mx[2] - (double)mn[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    int max_axis() const {                                                              // aabb.ts:50-64
        double x = (double)mx[0] - (double)mn[0];
        double y = (double)mx[1] - (double)mn[1];
        double z = (double)mx[2] - (double)mn[2];
        if (x > y && x > z) return 0;
        if (y > x && y > z) return 1;
        return 2;
    }
};

Box range_box(const ptmi_triangle *t, uint32_t s, uint32_t e) {
    Box b; b.reset();
    for (uint32_t i = s; i < e; i++) b.grow(t[i]);
    return b;
}

struct Keyed { double key; uint32_t idx; };
struct CmpKeyed { double operator()(const Keyed &x, const Keyed &y) const { return x.key - y.key; } };

void put_node(ptmi_bvh_node &n, const Box &b, uint32_t off, uint32_t cnt) {
    std::memset(&n, 0, sizeof n);
    for (int k = 0; k < 3; k++) { n.aabb_min[k] = b.mn[k]; n.aabb_max[k] = b.mx[k]; }
    n.left = 0xFFFFFFFFu; n.right = 0xFFFFFFFFu;          // -1 through Uint32Array
    n.triangle_offset = off; n.triangle_count = cnt;
}


// One node of bvh.ts:96-131: sort the range along the longest axis and pick the cheapest of the `bins - 1`
// equal-count candidates. Returns the split position, or `start` when no candidate has a finite cost.
struct SplitScratch { std::vector<Keyed> keys; std::vector<ptmi_triangle> tmp; std::vector<uint32_t> cand; std::vector<Box> lbox, rbox; };

uint32_t choose_split(ptmi_triangle *tris, uint32_t start, uint32_t end, uint32_t bins, SplitScratch &w) {
    const uint32_t num = end - start;
    int axis = range_box(tris, start, end).max_axis();                                  // bvh.ts:96-97
    // bvh.ts:100-102, :160-168: sort the range by (v0+v1+v2)[axis] / 3
    w.keys.resize(num);
    for (uint32_t i = 0; i < num; i++) {
        const ptmi_triangle &q = tris[start + i];
        w.keys[i].key = ((double)q.v0[axis] + (double)q.v1[axis] + (double)q.v2[axis]) / 3.0;
        w.keys[i].idx = start + i;
    }
    PartialSorter<Keyed, CmpKeyed> sorter{w.keys.data(), CmpKeyed()};
    sorter.run(0, (int64_t)num);
    w.tmp.resize(num);
    for (uint32_t i = 0; i < num; i++) w.tmp[i] = tris[w.keys[i].idx];
    std::memcpy(tris + start, w.tmp.data(), (size_t)num * sizeof(ptmi_triangle));

    // bvh.ts:171-199: candidates at start + floor(num * i/bins); left/right boxes from one forward and one
    // backward sweep (min/max are exact, so the boxes equal the reference's per-candidate recomputation)
    w.cand.clear();
    for (uint32_t i = 1; i < bins; i++) {
        double ratio = (double)i / (double)bins;
        uint32_t split = start + (uint32_t)std::floor((double)num * ratio);
        if (split == start || split == end) continue;
        w.cand.push_back(split);
    }
    w.lbox.resize(w.cand.size()); w.rbox.resize(w.cand.size());
    {
        Box b; b.reset(); uint32_t pos = start;
        for (size_t c = 0; c < w.cand.size(); c++) {
            for (; pos < w.cand[c]; pos++) b.grow(tris[pos]);
            w.lbox[c] = b;
        }
        b.reset(); pos = end;
        for (size_t c = w.cand.size(); c-- > 0;) {
            for (; pos > w.cand[c]; pos--) b.grow(tris[pos - 1]);
            w.rbox[c] = b;
        }
    }
    double min_cost = std::numeric_limits<double>::infinity();
    uint32_t best = start;
    for (size_t c = 0; c < w.cand.size(); c++) {
        double lc = w.lbox[c].area() * (double)(w.cand[c] - start);
        double rc = w.rbox[c].area() * (double)(end - w.cand[c]);
        double cost = 1.0 + (lc + rc) * 2.0;                                            // bvh.ts:206-229
        if (cost < min_cost) { min_cost = cost; best = w.cand[c]; }
    }
    return best;
}

// ---- the same tree from several threads ------------------------------------------------------------------
// The reference's explicit stack (bvh.ts:74-152) pops the right child first, so the nodes below a split are
// numbered: left, right, then everything under right, then everything under left. Disjoint triangle ranges are
// independent, so the two sides can be built concurrently into their own arrays (links relative to the array)
// and spliced in that order; the result is byte-identical to the one-thread loop.
constexpr uint32_t kParallelMin = 1u << 15;      // ranges below this are built by the calling thread
std::atomic<int> g_threads{0};                    // 0 = one per hardware thread (at most 32)

struct Subtree {
    std::vector<ptmi_bvh_node> below;             // all nodes under the range's own node, in the reference's order
    uint32_t depth = 0;                           // deepest leaf, counted from the root = 1
    bool failed = false;
};

void rebase(std::vector<ptmi_bvh_node> &v, uint32_t by) {
    for (ptmi_bvh_node &n : v)
        if (n.triangle_count == 0) { n.left += by; n.right += by; }
}

void build_below(ptmi_triangle *tris, uint32_t start, uint32_t end, uint32_t depth, uint32_t max_leaf, uint32_t bins,
                 std::atomic<int> &spare, Subtree &out) {
    if (end - start <= kParallelMin) {
        // the reference's loop, numbering relative to out.below
        struct Task { int64_t node; uint32_t start, end, depth; };   // node -1 = the range's own node (not in `below`)
        SplitScratch w;
        std::vector<Task> work;
        work.push_back({-1, start, end, depth});
        while (!work.empty()) {
            Task t = work.back(); work.pop_back();
            if (t.depth > out.depth) out.depth = t.depth;
            if (t.end - t.start <= max_leaf) continue;                // leaves keep what put_node wrote
            uint32_t best = choose_split(tris, t.start, t.end, bins, w);
            if (best == t.start) { out.failed = true; return; }
            uint32_t li = (uint32_t)out.below.size(), ri = li + 1;
            out.below.resize(out.below.size() + 2);
            put_node(out.below[li], range_box(tris, t.start, best), t.start, best - t.start);
            put_node(out.below[ri], range_box(tris, best, t.end), best, t.end - best);
            if (t.node >= 0) {
                ptmi_bvh_node &p = out.below[(size_t)t.node];
                p.left = li; p.right = ri; p.triangle_count = 0; p.triangle_offset = 0;
            }
            work.push_back({(int64_t)li, t.start, best, t.depth + 1});
            work.push_back({(int64_t)ri, best, t.end, t.depth + 1});
        }
        return;
    }
    if (depth > out.depth) out.depth = depth;
    SplitScratch w;
    uint32_t best = choose_split(tris, start, end, bins, w);
    if (best == start) { out.failed = true; return; }
    w = SplitScratch();
    Subtree L, R;
    bool spawned = false;
    std::thread th;
    int have = spare.load();
    while (have > 0 && !spare.compare_exchange_weak(have, have - 1)) {}
    if (have > 0) {
        spawned = true;
        th = std::thread([&] { build_below(tris, best, end, depth + 1, max_leaf, bins, spare, R); });
    } else {
        build_below(tris, best, end, depth + 1, max_leaf, bins, spare, R);
    }
    build_below(tris, start, best, depth + 1, max_leaf, bins, spare, L);
    if (spawned) { th.join(); spare.fetch_add(1); }
    if (L.failed || R.failed) { out.failed = true; return; }
    // below = [left, right] + below(right) + below(left); a side's own children are the first two of its array
    out.below.resize(2);
    put_node(out.below[0], range_box(tris, start, best), start, best - start);
    put_node(out.below[1], range_box(tris, best, end), best, end - best);
    const uint32_t r_at = 2, l_at = 2 + (uint32_t)R.below.size();
    if (!R.below.empty()) { out.below[1].left = r_at; out.below[1].right = r_at + 1; out.below[1].triangle_count = 0; out.below[1].triangle_offset = 0; }
    if (!L.below.empty()) { out.below[0].left = l_at; out.below[0].right = l_at + 1; out.below[0].triangle_count = 0; out.below[0].triangle_offset = 0; }
    rebase(R.below, r_at); rebase(L.below, l_at);
    out.below.insert(out.below.end(), R.below.begin(), R.below.end());
    out.below.insert(out.below.end(), L.below.begin(), L.below.end());
    out.depth = std::max(out.depth, std::max(L.depth, R.depth));
}

}  // namespace

extern "C" {

const char *ptmi_scene_last_error(void) { return g_err.c_str(); }

int ptmi_scene_sort_partially_f64(double *arr, int64_t n, int64_t start, int64_t end, int descending) {
    if (!arr) return fail(-2, "null array");
    if (start < 0 || end > n || start >= end) {                                         // arr.ts:7-10
        char buf[96];
        std::snprintf(buf, sizeof buf, "Invalid indices: start=%lld, end=%lld", (long long)start, (long long)end);
        return fail(-1, buf);
    }
    if (descending) { PartialSorter<double, CmpDesc> s{arr, CmpDesc()}; s.run(start, end); }
    else { PartialSorter<double, CmpAsc> s{arr, CmpAsc()}; s.run(start, end); }
    return 0;
}

uint32_t ptmi_scene_bvh_node_bound(uint32_t n_tris) { return n_tris ? 2u * n_tris - 1u : 1u; }

void ptmi_scene_set_threads(int n_threads) { g_threads.store(n_threads); }

int ptmi_scene_build_bvh(ptmi_triangle *tris, uint32_t n, uint32_t max_leaf, uint32_t bins,
                         ptmi_bvh_node *out, uint32_t cap, uint32_t *n_nodes_out,
                         uint32_t *max_depth_out) {
    if (!tris || !out || !n_nodes_out) return fail(-2, "null argument");
    if (max_leaf == 0) max_leaf = 4;                                                    // bvh.ts:86
    if (bins == 0) bins = 12;                                                           // bvh.ts:110
    if (cap < 1) return fail(-3, "node capacity too small");
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(tris[i].v0[k]) || !std::isfinite(tris[i].v1[k]) || !std::isfinite(tris[i].v2[k]))
                return fail(-4, "non-finite vertex position (the reference's builder does not terminate on it)");

    struct Task { uint32_t node, start, end, depth; };
    int threads = g_threads.load();
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads > 32) threads = 32; }
    if (threads > 1 && n > kParallelMin) {
        std::atomic<int> spare{threads - 1};
        Subtree sub;
        build_below(tris, 0, n, 1, max_leaf, bins, spare, sub);
        if (sub.failed) return fail(-5, "no finite SAH split (the reference's builder does not terminate here)");
        if (sub.below.size() + 1 > cap) return fail(-3, "node capacity too small");
        put_node(out[0], range_box(tris, 0, n), 0, n);                                  // bvh.ts:63-71
        if (!sub.below.empty()) { out[0].left = 1; out[0].right = 2; out[0].triangle_count = 0; out[0].triangle_offset = 0; }
        rebase(sub.below, 1);
        std::memcpy(out + 1, sub.below.data(), sub.below.size() * sizeof(ptmi_bvh_node));
        *n_nodes_out = (uint32_t)sub.below.size() + 1;
        if (max_depth_out) *max_depth_out = sub.depth;
        return 0;
    }

    std::vector<Task> work;
    SplitScratch scratch;
    uint32_t count = 0, max_depth = 0;

    put_node(out[count++], range_box(tris, 0, n), 0, n);                                // bvh.ts:63-71
    work.push_back({0u, 0u, n, 1u});
    while (!work.empty()) {
        Task t = work.back(); work.pop_back();
        uint32_t num = t.end - t.start;
        if (t.depth > max_depth) max_depth = t.depth;
        if (num <= max_leaf) {                                                          // bvh.ts:86-92
            out[t.node].left = out[t.node].right = 0xFFFFFFFFu;
            out[t.node].triangle_offset = t.start;
            out[t.node].triangle_count = num;
            continue;
        }
        uint32_t best = choose_split(tris, t.start, t.end, bins, scratch);
        if (best == t.start)
            return fail(-5, "no finite SAH split (the reference's builder does not terminate here)");
        if (count + 2 > cap) return fail(-3, "node capacity too small");
        uint32_t li = count, ri = count + 1;                                            // bvh.ts:113-134
        put_node(out[li], range_box(tris, t.start, best), t.start, best - t.start);
        put_node(out[ri], range_box(tris, best, t.end), best, t.end - best);
        count += 2;
        out[t.node].left = li; out[t.node].right = ri;
        out[t.node].triangle_count = 0; out[t.node].triangle_offset = 0;                // bvh.ts:137-138
        work.push_back({li, t.start, best, t.depth + 1});                               // bvh.ts:141-151
        work.push_back({ri, best, t.end, t.depth + 1});
    }
    *n_nodes_out = count;
    if (max_depth_out) *max_depth_out = max_depth;
    return 0;
}

int ptmi_scene_emissive_lights(const ptmi_triangle *tris, uint32_t n_tris,
                               const ptmi_material *mats, uint32_t n_mats,
                               ptmi_light *lights, uint32_t cap, uint32_t *n_io) {
    if (!tris || !mats || !lights || !n_io) return fail(-2, "null argument");
    uint32_t n = *n_io;
    for (uint32_t i = 0; i < n_tris; i++) {
        uint32_t mi = tris[i].material_index;
        if (mi >= n_mats) return fail(-6, "triangle references a missing material");
        const ptmi_material &m = mats[mi];
        double len = std::sqrt((double)m.emission[0] * m.emission[0] + (double)m.emission[1] * m.emission[1] +
                               (double)m.emission[2] * m.emission[2]);
        if (len > 0.0) {                                                                // gpu.ts:126
            if (n >= cap) return fail(-3, "light capacity too small");
            ptmi_light &l = lights[n++];
            std::memset(&l, 0, sizeof l);
            l.light_type = PTMI_LIGHT_EMISSIVE;
            l.color[0] = m.emission[0]; l.color[1] = m.emission[1]; l.color[2] = m.emission[2];
            l.intensity = m.emissive_strength;
            l.triangle_index = i;
        }
    }
    *n_io = n;
    return 0;
}

}  // extern "C"
