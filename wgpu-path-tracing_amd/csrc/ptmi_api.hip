// ptmi_api.hip — the C ABI of include/ptmi.h: context, resource upload, the wavefront
// dispatch loop and the per-stage debug entry points.
//
// Replaces the host side of the reference's compute pass (src/renderer/renderer.ts:
// createBuffers :242-355, createBindGroups :368-381, updateCamera + dispatch :403-431).
#include "ptmi.h"
#include "pt_device.h"
#include "fast_tree.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_create_err;

struct EventPair { hipEvent_t a, b; int kind; };   // kind: 0 dispatch, 1 extend, 2 shade, 3 shadow, 4 raygen, 5 compaction, 6 accumulate
constexpr size_t kMaxPendingEvents = 4096;        // a caller that never synchronises (a preview loop) must not grow the list without bound

}  // namespace

// The buffers of the wavefront batch in flight, and the second stream that lets `shadow` run beside the next bounce.
struct Lane {
    size_t cap = 0;
    DevPaths paths{};
    float2 *hits = nullptr;
    DevShadow sh[2]{};                                 // shadow records, double-buffered by bounce parity (overlap)
    uint32_t *queue[2] = {nullptr, nullptr}, *sq[2] = {nullptr, nullptr};
    uint64_t *alive = nullptr, *shadowm = nullptr;
    size_t mask_words = 0;
    uint32_t *word_off = nullptr, *counts = nullptr;
    uint32_t *d_spill = nullptr;          // node-stack overflow of the global traversal variant (128 MiB on 256 CUs; first use)
    uint32_t *d_spill_side = nullptr;     // ... of the `shadow` kernel when it runs beside `extend`
    uint8_t *d_occ = nullptr;
    hipStream_t side = nullptr;           // `shadow` of bounce b beside the kernels of bounce b + 1
    hipEvent_t ev_ready = nullptr, ev_shadow[2] = {nullptr, nullptr};
};

struct ptmi_ctx {
    int device = 0, n_cu = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    Lane lane;
    mutable std::string err;
    bool alloc_oom = false;                            // the last failed batch allocation ran out of device memory
    ptmi_options opt{};

    // scene (bindings 1, 2, 4, 5, 6)
    void *d_tris = nullptr, *d_mats = nullptr, *d_lights = nullptr, *d_atlas = nullptr;
    float4 *d_wnodes = nullptr, *d_tripos = nullptr, *d_fast_wnodes = nullptr;
    float4 *d_own_tripos = nullptr, *d_leafbox = nullptr, *d_wnodes16 = nullptr, *d_ref_wnodes16 = nullptr;
    uint4 *d_qnodes16 = nullptr;
    DevScene *d_scene = nullptr;                       // sc in device memory (DevScene::self), rewritten whenever sc changes               // own leaves: leaf-ordered triangle images, per-triangle reference leaf boxes
    uint4 *d_qnodes = nullptr; uint32_t *d_leaf_stream = nullptr;        // quantised image of the rebuilt hierarchy (global variant)
    DevScene sc{};
    uint32_t bvh_depth = 0;
    uint32_t own_depth = 0;                  // own leaves: levels of the library's hierarchy (the uploaded tree's: bvh_depth)
    bool own_quant = false;                  // ... and whether it has a quantised image
    bool have_scene = false;
    size_t lds_scene_bytes = 0;

    // output (binding 0)
    uint32_t W = 0, H = 0;
    float4 *d_out_own = nullptr, *d_out = nullptr;

    unsigned long long *d_stats = nullptr;
    float4 *d_blit_f32 = nullptr; uint32_t *d_blit_u8 = nullptr; size_t blit_px = 0;   // canvas staging of ptmi_blit, kept between calls

    // statistics
    ptmi_stats st{};
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> event_pool;
    std::deque<hipEvent_t> in_flight;                  // one event per ptmi_dispatch, recorded behind its last kernel (ptmi_throttle)
};

namespace {

constexpr int kStatsWords = 8 + 64;
constexpr int kShadowCount = 72;          // slot of the shadow-queue length in ctx->counts (80 words)
constexpr size_t kLdsMax = 160 * 1024;
int fail(const ptmi_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}
#define HIP_TRY(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail((c), PTMI_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

template <class T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

void default_options(ptmi_options &o) {
    std::memset(&o, 0, sizeof o);
    o.max_bounces = 8; o.do_mis = 1; o.cull = 1; o.traversal = PTMI_TRAVERSAL_AUTO; o.overlap = 2;
}

hipEvent_t get_event(ptmi_ctx *c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}

// resolve finished event pairs into the statistics (stream must be synchronised)
void drain_events(ptmi_ctx *c) {
    for (auto &p : c->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            switch (p.kind) {
            case 0: c->st.gpu_ms += ms; break;
            case 1: c->st.extend_ms += ms; c->st.extend_launches++; break;
            case 2: c->st.shade_ms += ms; c->st.shade_launches++; break;
            case 3: c->st.shadow_ms += ms; c->st.shadow_launches++; break;
            case 4: c->st.raygen_ms += ms; break;
            case 5: c->st.compact_ms += ms; break;
            case 6: c->st.accumulate_ms += ms; break;
            }
        }
        c->event_pool.push_back(p.a); c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

// without a synchronisation: resolve the pairs at the front of the list whose closing event has completed
void drain_completed_events(ptmi_ctx *c) {
    size_t n = 0;
    while (n < c->pending.size() && hipEventQuery(c->pending[n].b) == hipSuccess) n++;
    if (n == 0) return;
    std::vector<EventPair> rest(c->pending.begin() + n, c->pending.end());
    c->pending.resize(n);
    drain_events(c);
    c->pending = std::move(rest);
}

constexpr size_t kMaxDispatchesInFlight = 256;     // a caller that never throttles or synchronises still cannot queue without bound

// drop the finished dispatches from the front of the list, then wait for the oldest ones until at most `max` are left
hipError_t throttle(ptmi_ctx *c, size_t max) {
    while (!c->in_flight.empty() && hipEventQuery(c->in_flight.front()) == hipSuccess) {
        c->event_pool.push_back(c->in_flight.front()); c->in_flight.pop_front();
    }
    (void)hipGetLastError();                          // hipErrorNotReady of the query is not an error
    while (c->in_flight.size() > max) {
        hipError_t e = hipEventSynchronize(c->in_flight.front());
        if (e != hipSuccess) return e;
        c->event_pool.push_back(c->in_flight.front()); c->in_flight.pop_front();
    }
    return hipSuccess;
}

struct Timed {
    ptmi_ctx *c; hipEvent_t a = nullptr, b = nullptr; int kind; bool on;
    hipStream_t st;
    Timed(ptmi_ctx *c_, int kind_, bool on_, hipStream_t st_ = nullptr) : c(c_), kind(kind_), on(on_), st(st_ ? st_ : c_->stream) {
        if (on) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, st); }
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(b, st);
        c->pending.push_back({a, b, kind});
        if (c->pending.size() > kMaxPendingEvents) drain_completed_events(c);
    }
};

void free_batch(Lane &ln) {
    dfree(ln.paths.O); dfree(ln.paths.D); dfree(ln.paths.C); dfree(ln.paths.L);
    dfree(ln.hits);
    for (int k = 0; k < 2; k++) { dfree(ln.sh[k].SO); ln.sh[k].SD = nullptr; ln.sh[k].SC = nullptr; dfree(ln.sq[k]); }
    dfree(ln.queue[0]); dfree(ln.queue[1]); dfree(ln.alive); dfree(ln.shadowm); dfree(ln.word_off); dfree(ln.d_occ);
    ln.cap = 0;
}

// everything the library has in flight, on every stream it owns
hipError_t sync_all(ptmi_ctx *c) {
    hipError_t e = c->stream ? hipStreamSynchronize(c->stream) : hipSuccess;
    if (e == hipSuccess && c->lane.side) e = hipStreamSynchronize(c->lane.side);
    if (e == hipSuccess && c->stream) e = hipStreamSynchronize(c->stream);      // the accumulate that waited for the side stream
    return e;
}

// bytes of device memory a path of a batch takes in ensure_capacity (state 56 + hit 8 + 2 x (record 44 + index 4) + 2 queues + masks)
constexpr size_t kBytesPerPath = 16 + 16 + 8 + 16 + 8 + 2 * (16 + 16 + sizeof(rgb_sc) + 4) + 2 * 4 + 1 + 1;

int ensure_capacity(ptmi_ctx *c, Lane &ln, size_t n) {
    if (n <= ln.cap) return PTMI_OK;
    HIP_TRY(c, sync_all(c));
    free_batch(ln);
    size_t cap = (n + 1023) & ~(size_t)1023;
    size_t words = cap / 64 + 1;
    size_t tiles = cap / pt_compact_tile_slots() + 2;
    c->alloc_oom = false;
    // a failed allocation leaves the lane empty (not half-built) and the runtime's sticky error cleared; ptmi_dispatch retries
    // with a smaller batch when it chose the size itself
#define ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc(&(ptr), (bytes)); if (e_ != hipSuccess) { \
        c->alloc_oom = e_ == hipErrorOutOfMemory; free_batch(ln); (void)hipGetLastError(); \
        return fail(c, PTMI_E_HIP, "hipMalloc of %zu bytes for a batch of %zu paths failed: %s", (size_t)(bytes), cap, hipGetErrorString(e_)); } } while (0)
    ALLOC(ln.paths.O, cap * 16); ALLOC(ln.paths.D, cap * 16);
    ALLOC(ln.paths.C, cap * 8); ALLOC(ln.paths.L, cap * 16);      // room for either stride
    ALLOC(ln.hits, cap * 8);
    for (int k = 0; k < 2; k++) {
        ALLOC(ln.sh[k].SO, cap * (16 + 16 + sizeof(rgb_sc)));
        ln.sh[k].SD = ln.sh[k].SO + cap; ln.sh[k].SC = reinterpret_cast<rgb_sc *>(ln.sh[k].SO + 2 * cap); ln.sh[k].cap = (uint32_t)cap;
        ALLOC(ln.sq[k], cap * 4);
    }
    ALLOC(ln.queue[0], cap * 4); ALLOC(ln.queue[1], cap * 4);
    ALLOC(ln.alive, words * 8); ALLOC(ln.shadowm, words * 8); ln.mask_words = words;
    ALLOC(ln.word_off, 2 * tiles * 4);
    ALLOC(ln.d_occ, cap);
#undef ALLOC
    ln.cap = cap;
    return PTMI_OK;
}

// ---- scene validation and the traversal image ---------------------------------
struct Built {
    std::vector<float4> wnodes, tripos;      // reference-shaped image
    std::vector<float4> fast_wnodes;         // SAH tree over the reference's leaves (empty: not applicable)
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t root_ref = PT_REF_NONE, depth = 0;
    uint32_t fast_root = PT_REF_NONE, fast_depth = 0;
    double tree_ms = 0.0;                    // time spent in pt_build_fast_tree
    float tri_safe_dsum = 0.0f;              // DevScene::tri_safe_dsum
    std::vector<uint4> qnodes; std::vector<uint32_t> leaf_stream;     // quantised image (empty: none)
    float q_origin[3] = {0, 0, 0}, q_scale[3] = {0, 0, 0};
    uint32_t q_top = 0;                      // quantised nodes numbered breadth-first at the front (LDS-resident in the kernel)
    uint32_t max_leaf_tris = 0;
    bool gpu_tree = false;                   // the rebuilt hierarchy came from the device (ptmi_options.tree_builder = 2)
    // own leaves (ptmi_options.leaves = 2)
    bool own = false;
    PtOwnTree own_tree;
    std::vector<uint4> own_qnodes;           // quantised nodes of own_tree (empty: a 16-bit grid does not resolve this scene)
    std::vector<float4> leafbox;             // 2 float4 per triangle (original index): its reference leaf's box
    std::vector<float4> own_wnodes16, ref_wnodes16;   // the two hierarchies with 16-bit child references (empty: the scene is too large for them)
    std::vector<uint4> own_qnodes16;         // own_qnodes with 16-bit child references (empty: no quantised image, or too large)
    uint32_t own_root16 = PT_REF_NONE, ref_root16 = PT_REF_NONE;
};

// a copy of a wide-node image whose child references fit 16 bits: an internal node's index, or 0x8000 | (count - 1) << 12 | first
// triangle. false: some reference does not fit (more than 32 767 nodes, a leaf beyond triangle 4 095 or of more than 8 triangles)
bool compact_ref(uint32_t r, uint32_t &o) {
    if (r & PT_REF_LEAF) {
        const uint32_t first = r & PT_LEAF_OFF_MASK, cnt = ((r >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u;
        if (first > 0xFFFu || cnt > 8u) return false;
        o = 0x8000u | ((cnt - 1u) << 12) | first;
    } else {
        if (r > 0x7FFFu) return false;
        o = r;
    }
    return true;
}
bool compact_refs(const std::vector<float4> &w, uint32_t root, std::vector<float4> &out, uint32_t &root16) {
    auto conv = compact_ref;
    out = w;
    if (root == PT_REF_NONE || !conv(root, root16)) return false;
    for (size_t i = 0; i < w.size() / 4; i++) {
        uint32_t l, r, l16, r16;
        std::memcpy(&l, &w[i * 4 + 3].x, 4); std::memcpy(&r, &w[i * 4 + 3].y, 4);
        if (!conv(l, l16) || !conv(r, r16)) return false;
        std::memcpy(&out[i * 4 + 3].x, &l16, 4); std::memcpy(&out[i * 4 + 3].y, &r16, 4);
    }
    return true;
}

#ifndef PT_LEAVES_DEFAULT
#define PT_LEAVES_DEFAULT 2            /* what ptmi_options.leaves = 0 means (measured: profiles/README.md) */
#endif
#ifndef PT_LEAF_TRIS_DEFAULT
#define PT_LEAF_TRIS_DEFAULT 2         /* ... and ptmi_options.leaf_tris = 0 */
#endif

uint32_t leaf_ref(const ptmi_bvh_node &n) {
    return PT_REF_LEAF | ((n.triangle_count - 1u) << PT_LEAF_OFF_BITS) | n.triangle_offset;
}

int build_image(ptmi_ctx *c, const ptmi_triangle *tris, uint32_t nt, const ptmi_bvh_node *nodes, uint32_t nn, Built &b) {
    if (nt == 0 || nn == 0) return PTMI_OK;                       // empty scene: every ray misses
    if (nt > PT_LEAF_OFF_MASK) return fail(c, PTMI_E_UNSUPPORTED, "more than %u triangles", PT_LEAF_OFF_MASK);
    // leaf <=> triangleCount > 0 (pt.wgsl:271)
    auto check_leaf = [&](uint32_t i) -> int {
        const ptmi_bvh_node &n = nodes[i];
        if (n.triangle_count > PT_LEAF_MAX_TRIS)
            return fail(c, PTMI_E_UNSUPPORTED, "BVH leaf %u holds %u triangles (limit %u)", i, n.triangle_count, PT_LEAF_MAX_TRIS);
        if ((uint64_t)n.triangle_offset + n.triangle_count > nt)
            return fail(c, PTMI_E_INVALID, "BVH leaf %u references triangles [%u,+%u) beyond %u", i, n.triangle_offset, n.triangle_count, nt);
        return PTMI_OK;
    };
    std::vector<uint32_t> wide_of(nn, PT_REF_NONE);
    std::vector<uint8_t> seen(nn, 0);
    struct Item { uint32_t node, depth; };
    std::vector<Item> stack;
    // pass 1: preorder (left first) numbering of the internal nodes
    stack.push_back({0u, 1u});
    uint32_t n_wide = 0;
    uint64_t next_offset = 0;               // leaves must come in ascending triangle order along the left-first DFS (below)
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.node >= nn) return fail(c, PTMI_E_INVALID, "BVH child index %u out of range (%u nodes)", it.node, nn);
        if (seen[it.node]) return fail(c, PTMI_E_INVALID, "BVH node %u is reachable twice", it.node);
        seen[it.node] = 1;
        b.depth = std::max(b.depth, it.depth);
        if (it.depth > 62) return fail(c, PTMI_E_UNSUPPORTED, "BVH deeper than 62 levels (the reference's own traversal stack holds 64 entries, pt.wgsl:249)");
        const ptmi_bvh_node &n = nodes[it.node];
        if (n.triangle_count > 0) {
            int rc = check_leaf(it.node); if (rc) return rc;
            // pt.wgsl:274 keeps the FIRST of equally near hits in its left-first DFS; the kernels visit leaves in another
            // order and break ties by the lowest triangle index. The two agree iff leaf ranges ascend along that DFS —
            // true of every tree bvh.ts builds (children split one contiguous range, left = lower part, bvh.ts:114-127).
            if (n.triangle_offset < next_offset)
                return fail(c, PTMI_E_UNSUPPORTED, "BVH leaf %u starts at triangle %u but an earlier leaf of the left-first DFS ends at %llu: "
                            "leaf ranges must ascend in DFS order (as bvh.ts builds them)", it.node, n.triangle_offset, (unsigned long long)next_offset);
            next_offset = (uint64_t)n.triangle_offset + n.triangle_count;
            b.max_leaf_tris = std::max(b.max_leaf_tris, n.triangle_count);
            continue;
        }
        wide_of[it.node] = n_wide++;
        stack.push_back({n.right, it.depth + 1});
        stack.push_back({n.left, it.depth + 1});
    }
    b.wnodes.assign((size_t)n_wide * 4, make_float4(0, 0, 0, 0));
    auto ref_of = [&](uint32_t i) { return nodes[i].triangle_count > 0 ? leaf_ref(nodes[i]) : wide_of[i]; };
    for (uint32_t i = 0; i < nn; i++) {
        if (wide_of[i] == PT_REF_NONE) continue;
        const ptmi_bvh_node &L = nodes[nodes[i].left], &R = nodes[nodes[i].right];
        float4 *w = &b.wnodes[(size_t)wide_of[i] * 4];
        w[0] = make_float4(L.aabb_min[0], L.aabb_min[1], L.aabb_min[2], L.aabb_max[0]);
        w[1] = make_float4(L.aabb_max[1], L.aabb_max[2], R.aabb_min[0], R.aabb_min[1]);
        w[2] = make_float4(R.aabb_min[2], R.aabb_max[0], R.aabb_max[1], R.aabb_max[2]);
        uint32_t lr = ref_of(nodes[i].left), rr = ref_of(nodes[i].right);
        float fl, fr; std::memcpy(&fl, &lr, 4); std::memcpy(&fr, &rr, 4);
        w[3] = make_float4(fl, fr, 0.0f, 0.0f);
    }
    for (int k = 0; k < 3; k++) { b.root_min[k] = nodes[0].aabb_min[k]; b.root_max[k] = nodes[0].aabb_max[k]; }
    b.root_ref = ref_of(0);
    // Nested tree (each node box contains its children's, all finite)? Then rebuild the hierarchy over the
    // reference's leaves (fast_tree.hip explains why the results cannot change).
    bool nested = n_wide > 0;
    std::vector<PtFastLeaf> leaves;
    for (uint32_t i = 0; i < nn && nested; i++) {
        if (!seen[i]) continue;
        const ptmi_bvh_node &n = nodes[i];
        for (int k = 0; k < 3; k++) nested = nested && std::isfinite(n.aabb_min[k]) && std::isfinite(n.aabb_max[k]);
        if (n.triangle_count > 0) {
            PtFastLeaf l;
            for (int k = 0; k < 3; k++) { l.mn[k] = n.aabb_min[k]; l.mx[k] = n.aabb_max[k]; }
            l.ref = leaf_ref(n); l.weight = n.triangle_count;
            leaves.push_back(l);
        } else {
            for (uint32_t ch : {n.left, n.right})
                for (int k = 0; k < 3; k++)
                    nested = nested && nodes[ch].aabb_min[k] >= n.aabb_min[k] && nodes[ch].aabb_max[k] <= n.aabb_max[k];
        }
    }
    const uint32_t leaves_mode = c->opt.leaves ? c->opt.leaves : (uint32_t)PT_LEAVES_DEFAULT;
    if (nested && leaves_mode == 2u && !c->opt.keep_reference_tree) {
        // The library's own leaves (fast_tree.h). What the reference's semantics need from the uploaded tree is kept beside them: the
        // tree itself (slow rays walk it) and, per triangle, the box of the leaf that lists it (the winner's verification).
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> which;
        which.reserve(nt);
        b.leafbox.assign((size_t)nt * 2, make_float4(0, 0, 0, 0));
        for (const PtFastLeaf &l : leaves) {            // (leaf ranges ascend and do not overlap: checked above)
            const uint32_t first = l.ref & PT_LEAF_OFF_MASK;
            for (uint32_t k = 0; k < l.weight; k++) {
                which.push_back(first + k);
                b.leafbox[2 * (size_t)(first + k)] = make_float4(l.mn[0], l.mn[1], l.mn[2], 0.0f);
                b.leafbox[2 * (size_t)(first + k) + 1] = make_float4(l.mx[0], l.mx[1], l.mx[2], 0.0f);
            }
        }
        std::sort(which.begin(), which.end());
        const uint32_t k_max = c->opt.leaf_tris ? c->opt.leaf_tris : (uint32_t)PT_LEAF_TRIS_DEFAULT;
        // small scenes: at most 14 levels, so that a lane's whole node stack fits the 15 LDS entries of two workgroups per CU
        const uint32_t limit = which.size() <= 2048 ? 14u : 60u;
        b.own = pt_build_own_tree(tris, which, k_max, limit, b.own_tree);
        if (b.own && nt <= 4096u) {                     // small scenes: both hierarchies once more with 16-bit child references
            if (!compact_refs(b.own_tree.wnodes, b.own_tree.root_ref, b.own_wnodes16, b.own_root16) ||
                !compact_refs(b.wnodes, b.root_ref, b.ref_wnodes16, b.ref_root16)) { b.own_wnodes16.clear(); b.ref_wnodes16.clear(); }
        }
        if (b.own) {
            float qo[3], qs[3];
            if (pt_quantize_nodes(b.own_tree.wnodes, b.own_qnodes, qo, qs, PT_QCACHE_NODES, b.q_top))
                for (int k = 0; k < 3; k++) { b.q_origin[k] = qo[k]; b.q_scale[k] = qs[k]; }
            else b.own_qnodes.clear();
            if (!b.own_qnodes.empty() && !b.own_wnodes16.empty()) {     // (the quantised nodes are renumbered: node 0 stays the root)
                b.own_qnodes16 = b.own_qnodes;
                for (uint4 &q : b.own_qnodes16) if (!compact_ref(q.w, q.w)) { b.own_qnodes16.clear(); break; }
            }
        } else {
            b.leafbox.clear();
        }
        b.tree_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (nested && leaves.size() >= 2 && !c->opt.keep_reference_tree && !b.own) {
        const auto t0 = std::chrono::steady_clock::now();
        // tree_builder = 2: on the device (gpu_tree.hip); the host builder when that is not wanted, not possible (ptmi_debug_image_stats
        // has no device) or refused
        bool built = false;
        if (c->opt.tree_builder == 2u && c->stream) built = pt_build_fast_tree_gpu(leaves, b.fast_wnodes, b.fast_root, b.fast_depth, c->stream);
        b.gpu_tree = built;
        if (!built) pt_build_fast_tree(leaves, b.fast_wnodes, b.fast_root, b.fast_depth);
        b.tree_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // triangle images: v0, e1 = v1 - v0, e2 = v2 - v0 (pt.wgsl:128-129; one IEEE subtraction each)
    b.tripos.resize((size_t)nt * 3);
    for (uint32_t i = 0; i < nt; i++) {
        const ptmi_triangle &t = tris[i];
        b.tripos[3 * (size_t)i + 0] = make_float4(t.v0[0], t.v0[1], t.v0[2], 0.0f);
        b.tripos[3 * (size_t)i + 1] = make_float4(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], 0.0f);
        b.tripos[3 * (size_t)i + 2] = make_float4(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], 0.0f);
    }
    {   // longest edge squared, in double; NaN / inf edges give 0 (no ray is "bounded" then)
        double emax2 = 0.0; bool finite = true;
        for (size_t k = 0; k < b.tripos.size(); k++) {
            if (k % 3 == 0) continue;
            const float4 &e = b.tripos[k];
            const double l2 = (double)e.x * e.x + (double)e.y * e.y + (double)e.z * e.z;
            if (!(l2 <= 1.7e308)) finite = false; else if (l2 > emax2) emax2 = l2;
        }
        const double k = !finite ? 0.0 : (emax2 > 0.0 ? std::ldexp(1.0, 98) / emax2 : 3.0e38);
        b.tri_safe_dsum = (float)(k < 3.0e38 ? k : 3.0e38);
    }
    {
        const auto t0 = std::chrono::steady_clock::now();
        if (!b.fast_wnodes.empty() && !pt_quantize_tree(leaves, b.fast_wnodes, b.tripos, b.qnodes, b.leaf_stream, b.q_origin, b.q_scale,
                                                         PT_QCACHE_NODES, b.q_top)) {
            b.qnodes.clear(); b.leaf_stream.clear();
        }
        b.tree_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PTMI_OK;
}

// per-lane LDS entries: the node stack (<= depth - 2 deferred siblings) plus room for filed leaves
int stack_entries_for(uint32_t depth) { return depth + 2 <= 16 ? 16 : depth + 2 <= 24 ? 24 : depth + 2 <= 32 ? 32 : 64; }

// Own leaves (traverse_own.hip): which memory variant a kernel runs as. Sizes: exact nodes 64 B, quantised 32 B, triangle images
// 48 B, 4 KB of LDS per stack entry of a 1024-thread workgroup. PTMI_OWN_EXTEND / PTMI_OWN_SHADOW (a PT_VARIANT_OWN_* number) override
// the choice where it fits — for same-box A/Bs, not for users.
TraverseConfig own_config(const ptmi_ctx *c, bool closest_hit) {
    TraverseConfig cfg{};
    cfg.cull = c->opt.cull ? 1 : 0;
    cfg.lds_scene_bytes = c->lds_scene_bytes;
    cfg.wgs_per_cu = 1;
    const uint32_t depth = std::max(c->own_depth, c->bvh_depth);      // slow rays walk the uploaded tree on the same stacks
    const size_t ne = (size_t)c->sc.n_wnodes * 64, nq = (size_t)c->sc.n_wnodes * 32, tb = (size_t)c->sc.n_own_tris * 48;
    const bool quant = c->own_quant;
    const int full_stack = depth + 2 <= 16 ? 16 : depth + 2 <= 32 ? 32 : 0;
    const size_t full_b = (size_t)full_stack * 4096, two_b = (size_t)15 * 4096, spill_b = (size_t)16 * 4096;
    const bool two_ok = depth + 1 <= 15;                              // the whole node stack in 15 entries
    // quantised nodes with compact references, two workgroups per CU: the 16-bit entries (2 KB each per workgroup) take what the nodes leave
    int q16_entries = (c->sc.qnodes16 && nq + 64 < kLdsMax / 2) ? std::min<int>(15, (int)((kLdsMax / 2 - 64 - nq) / 2048)) : 0;
    if (const char *e = std::getenv("PTMI_OWN_Q16_ENTRIES"))          // tests: a shorter stack than fits (more spills), never below 8 to pass `fits`
        q16_entries = std::min(q16_entries, std::max(8, std::atoi(e)));
    auto fits = [&](int variant, int wgs) -> bool {
        switch (variant) {
        case PT_VARIANT_OWN_LDS: return full_stack && ne + tb + full_b <= kLdsMax;
        case PT_VARIANT_OWN_QLDS: return quant && full_stack && nq + tb + full_b <= kLdsMax;
        case PT_VARIANT_OWN_LDS_NODES: return wgs == 2 ? two_ok && ne + two_b <= kLdsMax / 2 : ne + spill_b <= kLdsMax;
        case PT_VARIANT_OWN_QLDS_NODES: return quant && (wgs == 2 ? two_ok && nq + two_b <= kLdsMax / 2 : nq + spill_b <= kLdsMax);
        case PT_VARIANT_OWN_QGLOBAL: return quant;
        case PT_VARIANT_OWN_GLOBAL: return true;
        case PT_VARIANT_OWN_LDS16_NODES: return wgs == 2 && c->sc.wnodes16 != nullptr && two_ok && ne + two_b / 2 <= kLdsMax / 2;
        case PT_VARIANT_OWN_QLDS16_NODES: return wgs == 2 && quant && q16_entries >= 8;
        }
        return false;
    };
    auto take = [&](int variant, int wgs) {
        cfg.variant = variant; cfg.wgs_per_cu = wgs;
        const bool lds_full = variant == PT_VARIANT_OWN_LDS || variant == PT_VARIANT_OWN_QLDS;
        const bool global = variant == PT_VARIANT_OWN_QGLOBAL || variant == PT_VARIANT_OWN_GLOBAL;
        const bool q16 = variant == PT_VARIANT_OWN_QLDS16_NODES;
        cfg.stack_entries = q16 ? q16_entries : lds_full ? full_stack : wgs == 2 ? 15 : 16;      // (16-bit entries in the compact-reference variants)
        cfg.wants_spill = (global || q16 || (!lds_full && wgs == 1)) ? 1 : 0;
        cfg.quantized = (variant == PT_VARIANT_OWN_QLDS || variant == PT_VARIANT_OWN_QLDS_NODES || variant == PT_VARIANT_OWN_QGLOBAL || q16) ? 1 : 0;
    };
    const bool big = c->lds_scene_bytes > ((size_t)4 << 20);         // beyond an XCD's L2: the quantised nodes pay (traverse_config below)
    if (c->opt.traversal == PTMI_TRAVERSAL_GLOBAL) { take(quant ? PT_VARIANT_OWN_QGLOBAL : PT_VARIANT_OWN_GLOBAL, 1); return cfg; }
    if (c->opt.traversal == PTMI_TRAVERSAL_GLOBAL_EXACT) { take(PT_VARIANT_OWN_GLOBAL, 1); return cfg; }
    if (c->opt.traversal == PTMI_TRAVERSAL_LDS) {
        if (fits(PT_VARIANT_OWN_LDS, 1)) take(PT_VARIANT_OWN_LDS, 1);
        else if (fits(PT_VARIANT_OWN_QLDS, 1)) take(PT_VARIANT_OWN_QLDS, 1);
        else take(PT_VARIANT_OWN_GLOBAL, 1);                          // the caller reports that it does not fit
        return cfg;
    }
    if (const char *e = std::getenv(closest_hit ? "PTMI_OWN_EXTEND" : "PTMI_OWN_SHADOW")) {
        const int raw = std::atoi(e);
        // e.g. 7 = quantised nodes, one workgroup; 17 = two; 20 = exact nodes with compact references; 21 = quantised nodes with compact references
        const int v = raw == 20 ? (int)PT_VARIANT_OWN_LDS16_NODES : raw == 21 ? (int)PT_VARIANT_OWN_QLDS16_NODES : raw % 10, w = raw >= 10 ? 2 : 1;
        if (fits(v, w)) { take(v, w); return cfg; }
    }
    struct Pick { int variant, wgs; };
    // Both kernels are box-step heavy over own leaves (7 - 8 dependent node fetches per ray against 3 - 4 triangle tests) and gain from
    // the second workgroup per CU — 8 waves per SIMD to cover them — more than from resident triangles (config 1, same box: any-hit
    // kernel from two workgroups with quantised nodes 17.1 ms beside the main stream against 21.1 from the full image, +2 % overall)
    static const Pick closest[] = {{PT_VARIANT_OWN_LDS_NODES, 2}, {PT_VARIANT_OWN_LDS16_NODES, 2}, {PT_VARIANT_OWN_QLDS_NODES, 2}, {PT_VARIANT_OWN_QLDS16_NODES, 2},
                                   {PT_VARIANT_OWN_LDS, 1}, {PT_VARIANT_OWN_QLDS, 1}, {PT_VARIANT_OWN_QLDS_NODES, 1}, {PT_VARIANT_OWN_LDS_NODES, 1}};
    static const Pick any[] = {{PT_VARIANT_OWN_LDS_NODES, 2}, {PT_VARIANT_OWN_LDS16_NODES, 2}, {PT_VARIANT_OWN_QLDS_NODES, 2}, {PT_VARIANT_OWN_QLDS16_NODES, 2},
                               {PT_VARIANT_OWN_LDS, 1}, {PT_VARIANT_OWN_QLDS, 1}, {PT_VARIANT_OWN_QLDS_NODES, 1}, {PT_VARIANT_OWN_LDS_NODES, 1}};
    if (!big) {
        if (closest_hit) { for (const Pick &p : closest) if (fits(p.variant, p.wgs)) { take(p.variant, p.wgs); return cfg; } }
        else for (const Pick &p : any) if (fits(p.variant, p.wgs)) { take(p.variant, p.wgs); return cfg; }
    }
    take((quant && big) ? PT_VARIANT_OWN_QGLOBAL : PT_VARIANT_OWN_GLOBAL, 1);
    return cfg;
}

// closest_hit: the extend kernel may take the node-cache variant (two workgroups per CU) when it fits
TraverseConfig traverse_config(const ptmi_ctx *c, bool closest_hit) {
    if (c->sc.own) return own_config(c, closest_hit);
    TraverseConfig cfg{};
    cfg.stack_entries = stack_entries_for(c->bvh_depth);
    cfg.cull = c->opt.cull ? 1 : 0;
    cfg.lds_scene_bytes = c->lds_scene_bytes;
    const bool have = c->sc.root_ref != PT_REF_NONE;
    const int lds_stack = cfg.stack_entries <= 16 ? 16 : 32;              // the sizes the LDS kernels are built for
    const bool fits = have && cfg.stack_entries <= 32 && c->lds_scene_bytes + (size_t)lds_stack * 1024 * 4 <= kLdsMax;
    const int small_stack = c->bvh_depth + 1 <= 15 ? 15 : 16;  // node stack <= depth - 2, plus >= 3 entries for filed leaves
    const bool node_cache = have && c->bvh_depth + 2 <= 16 &&
                            (size_t)c->sc.n_wnodes * 64 + (size_t)small_stack * 1024 * 4 <= kLdsMax / 2;
    cfg.spill = nullptr; cfg.wants_spill = 0; cfg.wgs_per_cu = 2;
    // The quantised image pays where node fetches leave the L2 (measured: the 1 M-triangle scene, 67 MB, extend -16 %); a scene
    // that an XCD's 4 MiB L2 holds is bound by the ALUs, and decoding costs more than the bytes save (cornell_spheres walked
    // from global memory: shadow +30 %). AUTO decides by size; GLOBAL asks for the quantised image, GLOBAL_EXACT for the exact one.
    cfg.quantized = c->opt.traversal == PTMI_TRAVERSAL_GLOBAL ||
                    (c->opt.traversal == PTMI_TRAVERSAL_AUTO && c->lds_scene_bytes > ((size_t)4 << 20));
    if (c->opt.traversal == PTMI_TRAVERSAL_GLOBAL || c->opt.traversal == PTMI_TRAVERSAL_GLOBAL_EXACT) cfg.variant = PT_VARIANT_GLOBAL;
#ifndef PT_SHADOW_NODE_CACHE
#define PT_SHADOW_NODE_CACHE 0
#endif
    // The any-hit kernel keeps the full LDS image, one workgroup per CU. From the node cache with two workgroups (80 scalar
    // registers since round 2) it is as fast by itself (8.53 ms per 64 spp either way) but takes every wave slot of its CUs: beside it
    // `shade` stretches from 16.5 to 18.3 ms and config 1 loses 4 % (9 767 -> 9 344); with one workgroup it is 40 % slower itself.
    else if ((closest_hit || PT_SHADOW_NODE_CACHE) && node_cache && c->opt.traversal == PTMI_TRAVERSAL_AUTO) {
        cfg.variant = PT_VARIANT_LDS_NODES; cfg.stack_entries = small_stack;
    } else if (fits) {
        cfg.variant = PT_VARIANT_LDS; cfg.stack_entries = lds_stack;
    }
    else if (have && closest_hit && c->opt.traversal == PTMI_TRAVERSAL_AUTO &&
             (size_t)c->sc.n_wnodes * 64 + (size_t)16 * 1024 * 4 <= kLdsMax) {
        // mid-size trees (up to 1536 wide nodes): all nodes in LDS, one workgroup per CU, stacks spill. Measured on
        // cornell_spheres against the global variant: extend -6 %, shadow +5 % (so closest hit only)
        cfg.variant = PT_VARIANT_LDS_NODES; cfg.wgs_per_cu = 1; cfg.stack_entries = 16; cfg.wants_spill = 1;
    }
    else cfg.variant = PT_VARIANT_GLOBAL;
    if (cfg.variant == PT_VARIANT_GLOBAL) { cfg.stack_entries = 16; cfg.wants_spill = 1; }   // deeper stacks spill
    return cfg;
}

// the radiance sits at 16-byte stride beside kernels that wait on node fetches from memory (pt_device.h DevPaths)
bool walks_memory_quantised(const TraverseConfig &cfg) {
    return cfg.quantized && (cfg.variant == PT_VARIANT_GLOBAL || cfg.variant == PT_VARIANT_OWN_QGLOBAL);
}

int check_ready(ptmi_ctx *c, bool need_output) {
    if (!c) return PTMI_E_INVALID;
    if (!c->have_scene) return fail(c, PTMI_E_STATE, "no scene uploaded (ptmi_upload_scene)");
    if (need_output && (!c->d_out || c->W == 0 || c->H == 0)) return fail(c, PTMI_E_STATE, "no output buffer (ptmi_resize)");
    return PTMI_OK;
}

int upload_rays(ptmi_ctx *c, uint32_t n, const float *o3, const float *d3, const float *w, float4 *dO, float4 *dD) {
    std::vector<float4> o(n), d(n);
    for (uint32_t i = 0; i < n; i++) {
        o[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], w ? w[i] : 0.0f);
        d[i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], 0.0f);
    }
    HIP_TRY(c, hipMemcpyAsync(dO, o.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, sync_all(c));
    return PTMI_OK;
}

}  // namespace

// rows of a context: all of [y0, y1), or its strips part, part + parts, ... (the last strip may be short)
DevBand pt_band_of(const ptmi_options &opt, uint32_t W, uint32_t H) {
    DevBand band{W, H, opt.tile_y0, opt.tile_y1 ? std::min(opt.tile_y1, H) : H,
                 std::max(1u, opt.tile_strip), std::max(1u, opt.tile_parts), opt.tile_part, 0u};
    if (band.y0 >= band.y1) return band;
    const uint32_t range = band.y1 - band.y0;
    if (band.parts <= 1u) band.rows = range;
    else
        for (uint32_t s0 = band.part * band.strip; s0 < range; s0 += band.parts * band.strip)
            band.rows += std::min(band.strip, range - s0);
    return band;
}
hipStream_t pt_ctx_stream(ptmi_ctx *c) { return c->stream; }
float4 *pt_ctx_output(ptmi_ctx *c) { return c->d_out; }
int pt_ctx_device(const ptmi_ctx *c) { return c->device; }
int pt_ctx_cus(const ptmi_ctx *c) { return c->n_cu; }

extern "C" {

int ptmi_abi_version(void) { return PTMI_ABI_VERSION; }

const char *ptmi_last_error(const ptmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int ptmi_create(int device_ordinal, ptmi_ctx **out) {
    if (!out) return fail(nullptr, PTMI_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PTMI_E_NODEVICE, "no HIP device available (%s); this library has no CPU backend",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, PTMI_E_INVALID, "device ordinal %d out of range (%d devices)", device_ordinal, n);
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PTMI_E_NODEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only",
                    device_ordinal, prop.gcnArchName);
    ptmi_ctx *c = new ptmi_ctx();
    c->device = device_ordinal;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    default_options(c->opt);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; return fail(nullptr, PTMI_E_HIP, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    {
        // The shadow stream has a priority level of its own — HIGH — because a priority level has hardware queues of its own. At the
        // caller's (normal) priority the runtime multiplexes it with every other normal-priority stream of the process onto a few
        // hardware queues, and what it gets depends on what was created before: the first context of a process is fine, a context made
        // after another one was destroyed got its shadow stream onto that one's old main queue and ran 6 - 9 % slower, at the
        // one-stream rate (tools/two_contexts.py b, profiles/r03_queues/). At high or at low priority that case is gone; in the ordinary
        // case the three are level (config 1: normal 9 966, high 9 970, low 9 939; config 3: 5 027 / 5 037 / 5 031, interleaved;
        // round 2 had measured low 0.8 - 1.3 % behind normal). -DPT_SIDE_NORMAL_PRIORITY / -DPT_SIDE_LOW_PRIORITY build the others.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        bool ok = true;
        Lane &ln = c->lane;
#if defined(PT_SIDE_LOW_PRIORITY)
        ok = ok && hipStreamCreateWithPriority(&ln.side, hipStreamNonBlocking, lo) == hipSuccess;
#elif defined(PT_SIDE_NORMAL_PRIORITY)
        ok = ok && hipStreamCreateWithFlags(&ln.side, hipStreamNonBlocking) == hipSuccess;
#else
        ok = ok && hipStreamCreateWithPriority(&ln.side, hipStreamNonBlocking, hi) == hipSuccess;
#endif
        for (hipEvent_t *e : {&ln.ev_ready, &ln.ev_shadow[0], &ln.ev_shadow[1]})
            ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipMalloc(&ln.counts, 80 * sizeof(uint32_t)) == hipSuccess;
        if (!ok) { ptmi_destroy(c); return fail(nullptr, PTMI_E_HIP, "stream / event creation failed"); }
    }
    if (hipMalloc(&c->d_stats, kStatsWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_scene, sizeof(DevScene)) != hipSuccess || hipMemset(c->d_scene, 0, sizeof(DevScene)) != hipSuccess ||
        hipMemset(c->d_stats, 0, kStatsWords * sizeof(unsigned long long)) != hipSuccess) {
        ptmi_destroy(c); return fail(nullptr, PTMI_E_HIP, "device allocation failed");
    }
    *out = c;
    return PTMI_OK;
}

int ptmi_destroy(ptmi_ctx *c) {
    if (!c) return PTMI_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    drain_events(c);
    for (hipEvent_t e : c->in_flight) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    {
        Lane &ln = c->lane;
        free_batch(ln);
        dfree(ln.counts); dfree(ln.d_spill); dfree(ln.d_spill_side);
        for (hipEvent_t e : {ln.ev_ready, ln.ev_shadow[0], ln.ev_shadow[1]}) if (e) (void)hipEventDestroy(e);
        if (ln.side) (void)hipStreamDestroy(ln.side);
    }
    dfree(c->d_tris); dfree(c->d_mats); dfree(c->d_lights); dfree(c->d_atlas); dfree(c->d_wnodes); dfree(c->d_tripos);
    dfree(c->d_fast_wnodes); dfree(c->d_qnodes); dfree(c->d_leaf_stream); dfree(c->d_own_tripos); dfree(c->d_leafbox); dfree(c->d_wnodes16); dfree(c->d_ref_wnodes16);
    dfree(c->d_qnodes16);
    dfree(c->d_out_own); dfree(c->d_stats); dfree(c->d_scene); dfree(c->d_blit_f32); dfree(c->d_blit_u8);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PTMI_OK;
}

}  // extern "C"

// A scene prepared on the host (validation + traversal image: everything of an upload that does not depend on the device), and the
// caller's blobs it was made from. ptmi_upload_scene = prepare + install; ptmi_multi_upload_scene prepares ONCE and installs on N devices.
struct PtPrepared {
    Built b;
    const ptmi_triangle *tris; uint32_t nt; const ptmi_material *mats; uint32_t nm;
    const ptmi_light *lights; uint32_t nl;
    double build_ms;
    ptmi_options opt;                        // what it was built under (leaves, leaf_tris, keep_reference_tree, tree_builder)
};

PtPrepared *pt_prepare_scene(ptmi_ctx *c, const ptmi_triangle *tris, uint32_t nt, const ptmi_material *mats, uint32_t nm,
                             const ptmi_bvh_node *nodes, uint32_t nn, const ptmi_light *lights, uint32_t nl, int *rc_out) {
    auto bad = [&](int rc) -> PtPrepared * { *rc_out = rc; return nullptr; };
    if (!c) return bad(PTMI_E_INVALID);
    if ((nt && !tris) || (nm && !mats) || (nn && !nodes) || (nl && !lights))
        return bad(fail(c, PTMI_E_INVALID, "NULL blob with a non-zero count"));
    if (hipSetDevice(c->device) != hipSuccess) return bad(fail(c, PTMI_E_HIP, "hipSetDevice(%d) failed", c->device));
    for (uint32_t i = 0; i < nl; i++) {
        if (lights[i].light_type > PTMI_LIGHT_POINT)
            return bad(fail(c, PTMI_E_INVALID, "light %u has unknown type %u", i, lights[i].light_type));
        if (lights[i].light_type == PTMI_LIGHT_EMISSIVE && lights[i].triangle_index >= nt)
            return bad(fail(c, PTMI_E_INVALID, "emissive light %u references triangle %u of %u", i, lights[i].triangle_index, nt));
    }
    const auto t_start = std::chrono::steady_clock::now();
    PtPrepared *p = new PtPrepared();
    int rc = build_image(c, tris, nt, nodes, nn, p->b);
    if (rc) { delete p; return bad(rc); }
    p->tris = tris; p->nt = nt; p->mats = mats; p->nm = nm; p->lights = lights; p->nl = nl;
    p->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    p->opt = c->opt;
    *rc_out = PTMI_OK;
    return p;
}
void pt_free_prepared(PtPrepared *p) { delete p; }

int pt_install_scene(ptmi_ctx *c, const PtPrepared *prep) {
    if (!c || !prep) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const auto t_start = clk::now();
    const Built &b = prep->b;
    const ptmi_triangle *tris = prep->tris; const uint32_t nt = prep->nt;
    const ptmi_material *mats = prep->mats; const uint32_t nm = prep->nm;
    const ptmi_light *lights = prep->lights; const uint32_t nl = prep->nl;
    // Allocate and fill the new buffers first; the context keeps its previous scene until all of them exist.
    const auto t_copy = clk::now();
    void *n_tris = nullptr, *n_mats = nullptr, *n_lights = nullptr;
    float4 *n_wnodes = nullptr, *n_tripos = nullptr, *n_fast = nullptr, *n_own_tripos = nullptr, *n_leafbox = nullptr;
    float4 *n_w16 = nullptr, *n_r16 = nullptr;
    const bool has16 = b.own && !b.own_wnodes16.empty();
    uint4 *n_qnodes = nullptr, *n_q16 = nullptr; uint32_t *n_stream = nullptr;
    const bool own = b.own;
    const bool hasq16 = has16 && !b.own_qnodes16.empty();
    const std::vector<uint4> &qn = own ? b.own_qnodes : b.qnodes;
    const bool quant = !qn.empty();
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        if (bytes == 0) { hipError_t e = hipMalloc(dst, 16); if (e != hipSuccess) return e; return hipMemset(*dst, 0, 16); }
        hipError_t e = hipMalloc(dst, bytes); if (e != hipSuccess) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    // n_fast: the hierarchy the regular rays walk when it is not the uploaded one — rebuilt over the reference's leaves, or the own tree
    const std::vector<float4> &walk = own ? b.own_tree.wnodes : b.fast_wnodes;
    const bool fast = own || !b.fast_wnodes.empty();
    hipError_t e = up(&n_tris, tris, (size_t)nt * sizeof(ptmi_triangle));
    if (e == hipSuccess) e = up(&n_mats, mats, (size_t)nm * sizeof(ptmi_material));
    if (e == hipSuccess) e = up(&n_lights, lights, (size_t)nl * sizeof(ptmi_light));
    if (e == hipSuccess) e = up(reinterpret_cast<void **>(&n_wnodes), b.wnodes.data(), b.wnodes.size() * 16);
    if (e == hipSuccess) e = up(reinterpret_cast<void **>(&n_tripos), b.tripos.data(), b.tripos.size() * 16);
    if (e == hipSuccess && fast) e = up(reinterpret_cast<void **>(&n_fast), walk.data(), walk.size() * 16);
    if (e == hipSuccess && own) e = up(reinterpret_cast<void **>(&n_own_tripos), b.own_tree.tripos.data(), b.own_tree.tripos.size() * 16);
    if (e == hipSuccess && own) e = up(reinterpret_cast<void **>(&n_leafbox), b.leafbox.data(), b.leafbox.size() * 16);
    if (e == hipSuccess && has16) e = up(reinterpret_cast<void **>(&n_w16), b.own_wnodes16.data(), b.own_wnodes16.size() * 16);
    if (e == hipSuccess && has16) e = up(reinterpret_cast<void **>(&n_r16), b.ref_wnodes16.data(), b.ref_wnodes16.size() * 16);
    if (e == hipSuccess && quant) e = up(reinterpret_cast<void **>(&n_qnodes), qn.data(), qn.size() * 16);
    if (e == hipSuccess && hasq16) e = up(reinterpret_cast<void **>(&n_q16), b.own_qnodes16.data(), b.own_qnodes16.size() * 16);
    if (e == hipSuccess && quant && !own) e = up(reinterpret_cast<void **>(&n_stream), b.leaf_stream.data(), b.leaf_stream.size() * 4);
    if (e != hipSuccess) {
        dfree(n_tris); dfree(n_mats); dfree(n_lights); dfree(n_wnodes); dfree(n_tripos); dfree(n_fast); dfree(n_qnodes); dfree(n_stream);
        dfree(n_own_tripos); dfree(n_leafbox); dfree(n_w16); dfree(n_r16); dfree(n_q16);
        return fail(c, PTMI_E_HIP, "scene upload failed: %s (the previous scene, if any, is still in place)", hipGetErrorString(e));
    }
    HIP_TRY(c, sync_all(c));                  // nothing in flight reads the old buffers any more
    dfree(c->d_tris); dfree(c->d_mats); dfree(c->d_lights); dfree(c->d_wnodes); dfree(c->d_tripos); dfree(c->d_fast_wnodes);
    dfree(c->d_qnodes); dfree(c->d_leaf_stream); dfree(c->d_own_tripos); dfree(c->d_leafbox); dfree(c->d_wnodes16); dfree(c->d_ref_wnodes16);
    c->d_wnodes16 = n_w16; c->d_ref_wnodes16 = n_r16;
    dfree(c->d_qnodes16); c->d_qnodes16 = n_q16;
    c->d_qnodes = n_qnodes; c->d_leaf_stream = n_stream;
    c->d_tris = n_tris; c->d_mats = n_mats; c->d_lights = n_lights;
    c->d_wnodes = n_wnodes; c->d_tripos = n_tripos; c->d_fast_wnodes = n_fast;
    c->d_own_tripos = n_own_tripos; c->d_leafbox = n_leafbox;
    DevScene &s = c->sc;
    s.tris = static_cast<const ptmi_triangle *>(c->d_tris); s.n_tris = nt;
    s.mats = static_cast<const ptmi_material *>(c->d_mats); s.n_mats = nm;
    s.lights = static_cast<const ptmi_light *>(c->d_lights); s.n_lights = nl;
    s.ref_wnodes = c->d_wnodes; s.ref_root_ref = b.root_ref; s.has_fast = fast ? 1u : 0u;
    s.wnodes = fast ? c->d_fast_wnodes : c->d_wnodes;
    s.n_wnodes = (uint32_t)((fast ? walk.size() : b.wnodes.size()) / 4);
    s.tripos = own ? c->d_own_tripos : c->d_tripos;
    s.ref_tripos = c->d_tripos;
    s.qnodes = c->d_qnodes; s.leaf_stream = c->d_leaf_stream;
    for (int k = 0; k < 3; k++) { s.q_origin[k] = b.q_origin[k]; s.q_scale[k] = b.q_scale[k]; }
    s.q_cached = b.q_top;
    s.tri_safe_dsum = b.tri_safe_dsum;
    for (int k = 0; k < 3; k++) {
        s.ref_root_min[k] = b.root_min[k]; s.ref_root_max[k] = b.root_max[k];
        s.root_min[k] = own ? b.own_tree.root_min[k] : b.root_min[k]; s.root_max[k] = own ? b.own_tree.root_max[k] : b.root_max[k];
    }
    s.root_ref = own ? b.own_tree.root_ref : fast ? b.fast_root : b.root_ref;
    s.own = own ? 1u : 0u;
    s.n_own_tris = own ? (uint32_t)(b.own_tree.tripos.size() / 3) : 0u;
    s.tri_leafbox = c->d_leafbox;
    s.wnodes16 = c->d_wnodes16; s.ref_wnodes16 = c->d_ref_wnodes16; s.qnodes16 = c->d_qnodes16;
    s.root_ref16 = has16 ? b.own_root16 : PT_REF_NONE; s.ref_root_ref16 = has16 ? b.ref_root16 : PT_REF_NONE;
    s.safe_origin = own ? b.own_tree.safe_origin : 0.0f;
    s.verify_stat = c->d_stats + 4;
    s.self = c->d_scene;
    HIP_TRY(c, hipMemcpy(c->d_scene, &c->sc, sizeof(DevScene), hipMemcpyHostToDevice));
    c->bvh_depth = std::max(b.depth, b.fast_depth);           // stacks must hold either tree (irregular rays use the uploaded one)
    c->own_depth = own ? b.own_tree.depth : 0u;
    c->own_quant = own && quant;
    c->lds_scene_bytes = (size_t)s.n_wnodes * 64 + (own ? b.own_tree.tripos.size() : b.tripos.size()) * 16;
    c->have_scene = true;
    c->st.leaves_used = own ? 2u : 1u;
    c->st.leaf_tris_used = own ? b.own_tree.max_leaf_tris : b.max_leaf_tris;
    c->st.upload_copy_ms = ms_since(t_copy);
    c->st.upload_tree_ms = b.tree_ms;
    c->st.upload_ms = prep->build_ms + ms_since(t_start);
    return PTMI_OK;
}

extern "C" {

int ptmi_upload_scene(ptmi_ctx *c, const ptmi_triangle *tris, uint32_t nt, const ptmi_material *mats, uint32_t nm,
                      const ptmi_bvh_node *nodes, uint32_t nn, const ptmi_light *lights, uint32_t nl) {
    int rc = PTMI_OK;
    PtPrepared *p = pt_prepare_scene(c, tris, nt, mats, nm, nodes, nn, lights, nl, &rc);
    if (!p) return rc;
    rc = pt_install_scene(c, p);
    pt_free_prepared(p);
    return rc;
}

int ptmi_upload_atlas(ptmi_ctx *c, const void *texels, uint32_t w, uint32_t h, int fmt) {
    if (!c) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    dfree(c->d_atlas);
    c->sc.atlas = nullptr; c->sc.atlas_w = c->sc.atlas_h = c->sc.atlas_fmt = 0;
    if (!texels || w == 0 || h == 0) return PTMI_OK;
    if (fmt != PTMI_ATLAS_RGBA16F && fmt != PTMI_ATLAS_RGBA32F) return fail(c, PTMI_E_INVALID, "unknown atlas format %d", fmt);
    size_t bytes = (size_t)w * h * (fmt == PTMI_ATLAS_RGBA16F ? 8 : 16);
    HIP_TRY(c, hipMalloc(&c->d_atlas, bytes));
    HIP_TRY(c, hipMemcpy(c->d_atlas, texels, bytes, hipMemcpyHostToDevice));
    c->sc.atlas = c->d_atlas; c->sc.atlas_w = w; c->sc.atlas_h = h; c->sc.atlas_fmt = (uint32_t)fmt;
    HIP_TRY(c, hipMemcpy(c->d_scene, &c->sc, sizeof(DevScene), hipMemcpyHostToDevice));
    return PTMI_OK;
}

int ptmi_resize(ptmi_ctx *c, uint32_t w, uint32_t h) {
    if (!c) return PTMI_E_INVALID;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(c, PTMI_E_INVALID, "bad size %ux%u", w, h);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    dfree(c->d_out_own);
    size_t bytes = (size_t)w * h * PTMI_OUTPUT_STRIDE;
    HIP_TRY(c, hipMalloc(&c->d_out_own, bytes));
    HIP_TRY(c, hipMemset(c->d_out_own, 0, bytes));
    c->d_out = c->d_out_own; c->W = w; c->H = h;
    return PTMI_OK;
}

int ptmi_set_options(ptmi_ctx *c, const ptmi_options *o) {
    if (!c || !o) return PTMI_E_INVALID;
    if (o->max_bounces < 1 || o->max_bounces > 64) return fail(c, PTMI_E_INVALID, "max_bounces %u not in 1..64", o->max_bounces);
    if (o->traversal > PTMI_TRAVERSAL_GLOBAL_EXACT) return fail(c, PTMI_E_INVALID, "unknown traversal mode %u", o->traversal);
    if (o->tile_y1 != 0 && o->tile_y0 >= o->tile_y1) return fail(c, PTMI_E_INVALID, "empty tile rows [%u,%u)", o->tile_y0, o->tile_y1);
    if (o->tile_parts > 1 && o->tile_part >= o->tile_parts)
        return fail(c, PTMI_E_INVALID, "tile_part %u is not below tile_parts %u", o->tile_part, o->tile_parts);
    if (o->perf_mode > 1) return fail(c, PTMI_E_INVALID, "unknown perf_mode %u", o->perf_mode);
    if (o->overlap > 2) return fail(c, PTMI_E_INVALID, "unknown overlap %u", o->overlap);
    if (o->reserved_a || o->reserved_b[0] || o->reserved_b[1] || o->reserved_b[2] || o->reserved_b[3] || o->reserved[0])
        return fail(c, PTMI_E_INVALID, "a reserved option word is not zero (ABI <= 3's ray_sort / worklist / tails / state / pipeline are gone: "
                    "start from ptmi_get_options)");
    if (o->leaves > 2) return fail(c, PTMI_E_INVALID, "unknown leaves %u", o->leaves);
    if (o->leaf_tris > PT_LEAF_MAX_TRIS) return fail(c, PTMI_E_INVALID, "leaf_tris %u above %u", o->leaf_tris, PT_LEAF_MAX_TRIS);
    if (o->tree_builder > 2) return fail(c, PTMI_E_INVALID, "unknown tree_builder %u", o->tree_builder);
    c->opt = *o;
    return PTMI_OK;
}
int ptmi_get_options(const ptmi_ctx *c, ptmi_options *o) {
    if (!c || !o) return PTMI_E_INVALID;
    *o = c->opt; return PTMI_OK;
}

int ptmi_dispatch(ptmi_ctx *c, const ptmi_camera *cam, uint32_t n_frames) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!cam) return fail(c, PTMI_E_INVALID, "camera is NULL");
    if (cam->width != c->W || cam->height != c->H)
        return fail(c, PTMI_E_INVALID, "camera says %ux%u but the output buffer is %ux%u", cam->width, cam->height, c->W, c->H);
    if (n_frames == 0) return PTMI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const DevBand band = pt_band_of(c->opt, c->W, c->H);
    if (band.y0 >= band.y1) return fail(c, PTMI_E_INVALID, "tile rows [%u,%u) outside the %u-row frame", band.y0, band.y1, c->H);
    if (band.rows == 0) return PTMI_OK;                         // more parts than strips: nothing to render here
    const uint64_t npix = (uint64_t)band.rows * band.width;
    uint32_t F = c->opt.frames_per_batch;
    // ~128 Mi paths, ~23 GB of state: the last bounces' small queues cost a fixed ~3 ms per batch, so fewer, larger batches
    // (measured at 1080p, Msamples/s: 32 frames 8 920, 64 frames 9 150 - 9 275, 128 frames 9 270 - 9 310)
    const bool auto_F = F == 0;
    Lane &ln = c->lane;
    if (auto_F) {
        F = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, (128ull << 20) / npix));
        // ... but never more than the device has room for: several contexts may share one device (ranks rehearsed on one GPU, a
        // Node host beside another process), and eight ranks of one node each size their batch by what THEIR device has free.
        // Room = free memory + what this context already holds, less a tenth for the rest (spill areas, blit staging).
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t held = (uint64_t)ln.cap * kBytesPerPath;
            const uint64_t room = (uint64_t)((double)(free_b + held) * 0.9);
            F = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(F, room / (npix * kBytesPerPath)));
        }
    }
    F = std::min(F, n_frames);
    const bool nee = c->opt.do_mis && c->sc.n_lights > 0;
    // overlap: `shadow` of bounce b on a side stream, beside extend / shade of bounce b + 1. It is then the only kernel that
    // adds to L (emissive hits leave a record too, ShadeParams::emit_records), bounce after bounce on one stream, so every
    // path's sum is formed in the same order as without it. Record buffers alternate by bounce parity; shade(b) waits for
    // shadow(b - 2), the end of the batch for the last one.
    const bool side = nee && c->opt.overlap != 0;
    if (npix * F > 0xFFFFFF00ull) return fail(c, PTMI_E_UNSUPPORTED, "batch of %llu paths exceeds 2^32", (unsigned long long)(npix * F));
    const TraverseConfig cfg0 = traverse_config(c, true), cfg_shadow0 = traverse_config(c, false);
    if (c->opt.traversal == PTMI_TRAVERSAL_LDS && cfg0.variant != PT_VARIANT_LDS && cfg0.variant != PT_VARIANT_OWN_LDS && cfg0.variant != PT_VARIANT_OWN_QLDS)
        return fail(c, PTMI_E_UNSUPPORTED, "scene needs %zu B of LDS plus the stack; it does not fit in %zu B", c->lds_scene_bytes, kLdsMax);
    for (;;) {
        rc = ensure_capacity(c, ln, (size_t)(npix * F));
        if (rc == PTMI_OK) break;
        // out of device memory with a batch size the library chose: halve it and try again (hipMemGetInfo is a snapshot; another
        // context may have allocated since). A size the caller asked for fails loudly.
        if (!auto_F || !c->alloc_oom || F <= 1) return rc;
        F = (F + 1) / 2;
    }
    if (cfg0.wants_spill && !ln.d_spill) HIP_TRY(c, hipMalloc(&ln.d_spill, pt_spill_bytes(c->n_cu * 8)));          // 128 MiB on 256 CUs
    if (cfg_shadow0.wants_spill && !ln.d_spill_side) HIP_TRY(c, hipMalloc(&ln.d_spill_side, pt_spill_bytes(c->n_cu * 8)));
#ifndef PT_L_STRIDE
#define PT_L_STRIDE 0              /* 0: by scene (pt_device.h, DevPaths); 3 or 4 floats: fixed */
#endif
    const bool from_memory = cfg0.variant == PT_VARIANT_GLOBAL || cfg0.variant == PT_VARIANT_OWN_QGLOBAL || cfg0.variant == PT_VARIANT_OWN_GLOBAL;
    c->st.traversal_used = from_memory ? PTMI_TRAVERSAL_GLOBAL : PTMI_TRAVERSAL_LDS;
    c->st.extend_variant = (uint32_t)cfg0.variant * 10u + (uint32_t)cfg0.wgs_per_cu;
    c->st.shadow_variant = (uint32_t)cfg_shadow0.variant * 10u + (uint32_t)cfg_shadow0.wgs_per_cu;
    c->st.frames_per_batch_used = F;
    c->st.radiance_stride_bytes = 4u * (PT_L_STRIDE ? (uint32_t)PT_L_STRIDE : ((walks_memory_quantised(cfg0) || walks_memory_quantised(cfg_shadow0)) ? 4u : 3u));
    const int blocks = c->n_cu * 8;
#ifndef PT_SHADE_WGS_PER_CU
#define PT_SHADE_WGS_PER_CU 16
#endif
    // 256-thread workgroups of the grid-stride shade kernel. Config 1, five interleaved runs each (Msamples/s): 8 per CU 9 247,
    // 16: 9 362, 32: 9 303, 64: 8 929 (run-to-run +-130); config 3 +-0.
    const int shade_blocks = c->n_cu * PT_SHADE_WGS_PER_CU;
    const uint32_t maxb = c->opt.max_bounces;
    const bool t1 = c->opt.timing >= 1, t2 = c->opt.timing >= 2, t3 = c->opt.timing >= 3;
    {
        Timed td(c, 0, t1);
        const hipStream_t ms = c->stream;                                         // the bounce loop's stream
        const hipStream_t ss = side ? ln.side : ms;                               // ... and the shadow kernels'
        const int tiles = (int)(ln.cap / pt_compact_tile_slots() + 1);
        TraverseConfig cfg = cfg0, cfg_shadow = cfg_shadow0;
        cfg.spill = ln.d_spill; cfg_shadow.spill = side ? ln.d_spill_side : ln.d_spill;
        if (cfg_shadow.wants_spill && !cfg_shadow.spill) cfg_shadow.spill = ln.d_spill_side;
        ln.paths.l_stride = c->st.radiance_stride_bytes / 4u;
        const DevPaths bp = ln.paths;
        for (uint32_t f0 = 0; f0 < n_frames; f0 += F) {
            const uint32_t fb = std::min(F, n_frames - f0);
            const uint32_t frame0 = cam->frame_index + f0;
            { Timed t(c, 4, t3, ms); pt_launch_raygen(ms, blocks, *cam, band, frame0, fb, bp, &ln.counts[0]); }
            int cur = 0;
            for (uint32_t b = 0; b < maxb; b++) {
                const uint32_t *q = b == 0 ? nullptr : ln.queue[cur];      // bounce 0: slot i holds path i
                const int par = side ? (int)(b & 1u) : 0;
                const ShadeParams shp{b, maxb, c->opt.do_mis, c->d_stats, side ? 1u : 0u};
                { Timed t(c, 1, t2, ms); (c->sc.own ? pt_launch_extend_own : pt_launch_extend)(ms, blocks, cfg, c->sc, bp, q, &ln.counts[b], ln.hits); }
                const bool last = b + 1 == maxb;
                if (side && b >= 2) HIP_TRY(c, hipStreamWaitEvent(ms, ln.ev_shadow[par], 0));      // its records are read
                { Timed t(c, 2, t3, ms);
                  (c->opt.perf_mode ? pt_launch_shade_fast : pt_launch_shade)(
                      ms, shade_blocks, c->sc, bp, q, &ln.counts[b], ln.hits, ln.sh[par], ln.alive, ln.shadowm, shp); }
                { Timed t(c, 5, t3, ms);
                  pt_launch_compact(ms, tiles, q, &ln.counts[b], ln.alive, nee ? ln.shadowm : nullptr,
                                    ln.word_off, ln.queue[cur ^ 1], &ln.counts[b + 1], ln.sq[par], &ln.counts[kShadowCount + par],
                                    c->d_stats, b, last ? 0 : 1); }
                if (side) {
                    HIP_TRY(c, hipEventRecord(ln.ev_ready, ms));
                    HIP_TRY(c, hipStreamWaitEvent(ss, ln.ev_ready, 0));
                    { Timed t(c, 3, t3, ss);
                      (c->sc.own ? pt_launch_shadow_own : pt_launch_shadow)(ss, blocks, cfg_shadow, c->sc, bp, ln.sh[par], ln.sq[par],
                                                                            &ln.counts[kShadowCount + par], nullptr); }
                    HIP_TRY(c, hipEventRecord(ln.ev_shadow[par], ss));
                } else if (nee) {
                    Timed t(c, 3, t3, ms);
                    (c->sc.own ? pt_launch_shadow_own : pt_launch_shadow)(ms, blocks, cfg_shadow, c->sc, bp, ln.sh[0], ln.sq[0], &ln.counts[kShadowCount], nullptr);
                }
                cur ^= 1;
            }
            // all additions to L are in before it is folded
            if (side) {
                HIP_TRY(c, hipStreamWaitEvent(ms, ln.ev_shadow[(maxb - 1) & 1u], 0));
                if (maxb >= 2) HIP_TRY(c, hipStreamWaitEvent(ms, ln.ev_shadow[maxb & 1u], 0));
            }
            { Timed t(c, 6, t3, ms); pt_launch_accumulate(ms, blocks, band, frame0, fb, bp.L, bp.l_stride, c->d_out); }
        }
    }
    HIP_TRY(c, hipGetLastError());
    {   // the end of this dispatch on the context's stream (the fold of its last batch): what ptmi_throttle waits for
        hipEvent_t done = get_event(c);
        HIP_TRY(c, hipEventRecord(done, c->stream));
        c->in_flight.push_back(done);
        if (c->in_flight.size() > kMaxDispatchesInFlight) HIP_TRY(c, throttle(c, kMaxDispatchesInFlight));
    }
    c->st.paths += npix * n_frames;
    c->st.frames += n_frames;
    c->st.dispatches += 1;
    return PTMI_OK;
}

int ptmi_throttle(ptmi_ctx *c, uint32_t max_in_flight, uint32_t *in_flight) {
    if (!c) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, throttle(c, max_in_flight));
    if (in_flight) *in_flight = (uint32_t)c->in_flight.size();
    return PTMI_OK;
}

int ptmi_synchronize(ptmi_ctx *c) {
    if (!c) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    HIP_TRY(c, throttle(c, 0));
    return PTMI_OK;
}

int ptmi_read_output(ptmi_ctx *c, float *dst, size_t n_floats) {
    if (!c || !dst) return PTMI_E_INVALID;
    if (!c->d_out) return fail(c, PTMI_E_STATE, "no output buffer (ptmi_resize)");
    if (n_floats != (size_t)c->W * c->H * 4) return fail(c, PTMI_E_INVALID, "expected %zu floats, got %zu", (size_t)c->W * c->H * 4, n_floats);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    HIP_TRY(c, hipMemcpy(dst, c->d_out, n_floats * 4, hipMemcpyDeviceToHost));
    return PTMI_OK;
}

int ptmi_write_output(ptmi_ctx *c, const float *src, size_t n_floats) {
    if (!c || !src) return PTMI_E_INVALID;
    if (!c->d_out) return fail(c, PTMI_E_STATE, "no output buffer (ptmi_resize)");
    if (n_floats != (size_t)c->W * c->H * 4) return fail(c, PTMI_E_INVALID, "expected %zu floats, got %zu", (size_t)c->W * c->H * 4, n_floats);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(c->d_out, src, n_floats * 4, hipMemcpyHostToDevice));
    return PTMI_OK;
}

void *ptmi_output_device_ptr(ptmi_ctx *c) { return c ? c->d_out : nullptr; }

int ptmi_bind_output_device(ptmi_ctx *c, void *p, size_t bytes) {
    if (!c) return PTMI_E_INVALID;
    if (c->W == 0) return fail(c, PTMI_E_STATE, "call ptmi_resize first");
    if (!p) { c->d_out = c->d_out_own; return PTMI_OK; }
    if (bytes < (size_t)c->W * c->H * PTMI_OUTPUT_STRIDE) return fail(c, PTMI_E_INVALID, "buffer of %zu bytes is too small", bytes);
    if (reinterpret_cast<uintptr_t>(p) & 15u) return fail(c, PTMI_E_INVALID, "buffer must be 16-byte aligned");
    c->d_out = static_cast<float4 *>(p);
    return PTMI_OK;
}

int ptmi_set_stream(ptmi_ctx *c, void *s) {
    if (!c) return PTMI_E_INVALID;
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
    return PTMI_OK;
}

int ptmi_blit(ptmi_ctx *c, float *dst_f32, size_t n_floats, uint8_t *dst_rgba8, size_t n_bytes) {
    if (!c) return PTMI_E_INVALID;
    if (!c->d_out) return fail(c, PTMI_E_STATE, "no output buffer (ptmi_resize)");
    if (!dst_f32 && !dst_rgba8) return PTMI_OK;
    const size_t n = (size_t)c->W * c->H;
    if (dst_f32 && n_floats != n * 4) return fail(c, PTMI_E_INVALID, "float canvas: expected %zu floats, got %zu", n * 4, n_floats);
    if (dst_rgba8 && n_bytes != n * 4) return fail(c, PTMI_E_INVALID, "8-bit canvas: expected %zu bytes, got %zu", n * 4, n_bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->blit_px != n) {                                       // staging buffers live in the context (one pair per size)
        HIP_TRY(c, sync_all(c));
        dfree(c->d_blit_f32); dfree(c->d_blit_u8); c->blit_px = 0;
    }
    if (dst_f32 && !c->d_blit_f32) HIP_TRY(c, hipMalloc(&c->d_blit_f32, n * 16));
    if (dst_rgba8 && !c->d_blit_u8) HIP_TRY(c, hipMalloc(&c->d_blit_u8, n * 4));
    c->blit_px = n;
    pt_launch_blit(c->stream, c->n_cu * 8, c->W, c->H, c->d_out, dst_f32 ? c->d_blit_f32 : nullptr, dst_rgba8 ? c->d_blit_u8 : nullptr);
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    if (dst_f32) HIP_TRY(c, hipMemcpy(dst_f32, c->d_blit_f32, n * 16, hipMemcpyDeviceToHost));
    if (dst_rgba8) HIP_TRY(c, hipMemcpy(dst_rgba8, c->d_blit_u8, n * 4, hipMemcpyDeviceToHost));
    return PTMI_OK;
}

int ptmi_get_size(const ptmi_ctx *c, uint32_t *w, uint32_t *h) {
    if (!c || !w || !h) return PTMI_E_INVALID;
    *w = c->W; *h = c->H;
    return PTMI_OK;
}

int ptmi_get_stats(ptmi_ctx *c, ptmi_stats *out) {
    if (!c || !out) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    unsigned long long h[kStatsWords];
    HIP_TRY(c, hipMemcpy(h, c->d_stats, sizeof h, hipMemcpyDeviceToHost));
    c->st.segments = h[0]; c->st.shadow_rays = h[1] - h[3]; c->st.shadow_traced = h[2] - h[3];      // h[3]: records of emissive hits
    for (int i = 0; i < 64; i++) c->st.segments_by_bounce[i] = h[8 + i];
    c->st.verify_failed = h[4];
    c->st.bvh_depth = c->bvh_depth;
    *out = c->st;
    return PTMI_OK;
}

int ptmi_reset_stats(ptmi_ctx *c) {
    if (!c) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    drain_events(c);
    HIP_TRY(c, hipMemset(c->d_stats, 0, kStatsWords * sizeof(unsigned long long)));
    const ptmi_stats old = c->st;
    std::memset(&c->st, 0, sizeof c->st);
    c->st.bvh_depth = c->bvh_depth;
    c->st.upload_ms = old.upload_ms; c->st.upload_tree_ms = old.upload_tree_ms; c->st.upload_copy_ms = old.upload_copy_ms;
    c->st.leaves_used = old.leaves_used; c->st.leaf_tris_used = old.leaf_tris_used;
    return PTMI_OK;
}

// ---- per-stage entry points ------------------------------------------------------
int ptmi_debug_raygen(ptmi_ctx *c, const ptmi_camera *cam, uint32_t n, const uint32_t *xs, const uint32_t *ys,
                      const uint32_t *frames, float *o3, float *d3, uint32_t *rng) {
    if (!c || !cam || !xs || !ys || !frames || !o3 || !d3) return PTMI_E_INVALID;
    if (n == 0) return PTMI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    Lane &ln = c->lane;                              // the per-stage entry points run on the context's stream
    HIP_TRY(c, sync_all(c));
    int rc = ensure_capacity(c, ln, n);
    if (rc) return rc;
    uint32_t *dx = ln.queue[0], *dy = ln.queue[1], *df = reinterpret_cast<uint32_t *>(ln.hits);
    HIP_TRY(c, hipMemcpyAsync(dx, xs, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dy, ys, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(df, frames, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    pt_launch_raygen_list(c->stream, *cam, n, dx, dy, df, ln.paths);
    std::vector<float4> o(n), d(n);
    HIP_TRY(c, hipMemcpyAsync(o.data(), ln.paths.O, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d.data(), ln.paths.D, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    for (uint32_t i = 0; i < n; i++) {
        o3[3 * i] = o[i].x; o3[3 * i + 1] = o[i].y; o3[3 * i + 2] = o[i].z;
        d3[3 * i] = d[i].x; d3[3 * i + 1] = d[i].y; d3[3 * i + 2] = d[i].z;
        if (rng) std::memcpy(&rng[i], &o[i].w, 4);
    }
    return PTMI_OK;
}

int ptmi_debug_intersect(ptmi_ctx *c, uint32_t n, const float *o3, const float *d3, float *t, uint32_t *tri,
                         float *u, float *v) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!o3 || !d3 || !t || !tri || !u || !v) return fail(c, PTMI_E_INVALID, "NULL argument");
    if (n == 0) return PTMI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    Lane &ln = c->lane;                              // the per-stage entry points run on the context's stream
    HIP_TRY(c, sync_all(c));
    rc = ensure_capacity(c, ln, n);
    if (rc) return rc;
    rc = upload_rays(c, n, o3, d3, nullptr, ln.paths.O, ln.paths.D);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(&ln.counts[0], &n, 4, hipMemcpyHostToDevice, c->stream));
    TraverseConfig cfg = traverse_config(c, true);
    if (c->opt.traversal == PTMI_TRAVERSAL_LDS && cfg.variant != PT_VARIANT_LDS && cfg.variant != PT_VARIANT_OWN_LDS && cfg.variant != PT_VARIANT_OWN_QLDS)
        return fail(c, PTMI_E_UNSUPPORTED, "scene does not fit in LDS");
    if (cfg.wants_spill && !ln.d_spill) HIP_TRY(c, hipMalloc(&ln.d_spill, pt_spill_bytes(c->n_cu * 8)));
    cfg.spill = ln.d_spill;
    c->st.extend_variant = (uint32_t)cfg.variant * 10u + (uint32_t)cfg.wgs_per_cu;
    (c->sc.own ? pt_launch_extend_own : pt_launch_extend)(c->stream, c->n_cu * 8, cfg, c->sc, ln.paths, nullptr, &ln.counts[0], ln.hits);
    // (u, v) are not part of the hit record: rebuilt exactly as `shade` rebuilds them (into the C stream, unused here)
    pt_launch_hit_uv(c->stream, n, c->sc, ln.paths, ln.hits, ln.paths.C);
    std::vector<float2> h(n), uv(n);
    HIP_TRY(c, hipMemcpyAsync(h.data(), ln.hits, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(uv.data(), ln.paths.C, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipGetLastError());
    for (uint32_t i = 0; i < n; i++) {
        t[i] = h[i].x; u[i] = uv[i].x; v[i] = uv[i].y; std::memcpy(&tri[i], &h[i].y, 4);
    }
    return PTMI_OK;
}

int ptmi_debug_occluded(ptmi_ctx *c, uint32_t n, const float *o3, const float *d3, const float *dist, uint8_t *occ) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!o3 || !d3 || !dist || !occ) return fail(c, PTMI_E_INVALID, "NULL argument");
    if (n == 0) return PTMI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    Lane &ln = c->lane;                              // the per-stage entry points run on the context's stream
    HIP_TRY(c, sync_all(c));
    rc = ensure_capacity(c, ln, n);
    if (rc) return rc;
    {   // every negative distance means "directional light" (ptmi.h). Inside the library -2 is the record of an emissive hit
        // (nothing to trace, traverse.hip ShadowIO::fetch): a caller's -2 must not be read as that, so negatives travel as -1
        std::vector<float> dn(dist, dist + n);
        for (float &x : dn) if (x < 0.0f) x = -1.0f;
        rc = upload_rays(c, n, o3, d3, dn.data(), ln.sh[0].SO, ln.sh[0].SD);
    }
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(&ln.counts[0], &n, 4, hipMemcpyHostToDevice, c->stream));
    TraverseConfig cfg = traverse_config(c, false);
    if (cfg.wants_spill && !ln.d_spill) HIP_TRY(c, hipMalloc(&ln.d_spill, pt_spill_bytes(c->n_cu * 8)));
    cfg.spill = ln.d_spill;
    c->st.shadow_variant = (uint32_t)cfg.variant * 10u + (uint32_t)cfg.wgs_per_cu;
    (c->sc.own ? pt_launch_shadow_own : pt_launch_shadow)(c->stream, c->n_cu * 8, cfg, c->sc, ln.paths, ln.sh[0], nullptr, &ln.counts[0], ln.d_occ);
    HIP_TRY(c, hipMemcpyAsync(occ, ln.d_occ, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipGetLastError());
    return PTMI_OK;
}

int ptmi_debug_image_stats(const ptmi_triangle *tris, uint32_t nt, const ptmi_bvh_node *nodes, uint32_t nn, double out[8]) {
    if (!out || (nt && !tris) || (nn && !nodes)) return PTMI_E_INVALID;
    for (int i = 0; i < 8; i++) out[i] = 0.0;
    ptmi_ctx tmp;                                   // host-only: never touches a device
    default_options(tmp.opt);
    tmp.opt.leaves = 1;                             // the image over the reference's leaves (ptmi_debug_build_image: the own one)
    Built b;
    int rc = build_image(&tmp, tris, nt, nodes, nn, b);
    if (rc) { g_create_err = tmp.err; return rc; }
    out[0] = (double)(b.fast_wnodes.size() / 4); out[2] = (double)b.fast_depth;
    out[3] = (double)(b.qnodes.size() / 2); out[4] = (double)b.leaf_stream.size();
    if (b.qnodes.empty()) return PTMI_OK;
    // every quantised child box, decoded with the kernel's own fmaf, must contain the exact child box it stands for
    double viol = 0.0, infl = 0.0; size_t boxes = 0, leaves = 0, bad_hdr = 0;
    auto area = [](const float *lo, const float *hi) {
        double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return 2.0 * (x * y + y * z + z * x);
    };
    // the quantised nodes are renumbered (top levels first): walk both images together from their roots
    std::vector<std::pair<uint32_t, uint32_t>> todo;       // (node of the exact image, node of the quantised image)
    todo.push_back({0u, 0u});
    size_t visited = 0;
    while (!todo.empty()) {
        const uint32_t i = todo.back().first, qi = todo.back().second;
        todo.pop_back();
        if ((size_t)qi * 2 + 1 >= b.qnodes.size() || (size_t)i * 4 + 3 >= b.fast_wnodes.size()) { bad_hdr++; continue; }
        visited++;
        const float4 *w = &b.fast_wnodes[(size_t)i * 4];
        const float lo[2][3] = {{w[0].x, w[0].y, w[0].z}, {w[1].z, w[1].w, w[2].x}};
        const float hi[2][3] = {{w[0].w, w[1].x, w[1].y}, {w[2].y, w[2].z, w[2].w}};
        uint32_t refs[2]; std::memcpy(&refs[0], &w[3].x, 4); std::memcpy(&refs[1], &w[3].y, 4);
        for (int ch = 0; ch < 2; ch++) {
            const uint4 q = b.qnodes[(size_t)qi * 2 + ch];
            const uint32_t pl[6] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16};   // lo.xyz, hi.xyz
            float dlo[3], dhi[3];
            for (int k = 0; k < 3; k++) {
                dlo[k] = std::fmaf(b.q_scale[k], (float)pl[k], b.q_origin[k]);
                dhi[k] = std::fmaf(b.q_scale[k], (float)pl[3 + k], b.q_origin[k]);
                if (!(dlo[k] <= lo[ch][k]) || !(dhi[k] >= hi[ch][k])) viol += 1.0;
            }
            const double a0 = area(lo[ch], hi[ch]);
            if (a0 > 0.0) { infl += area(dlo, dhi) / a0 - 1.0; boxes++; }
            if (refs[ch] & PT_REF_LEAF) {
                leaves++;
                if (!(q.w & PT_REF_LEAF)) { bad_hdr++; continue; }
                const uint32_t *h = &b.leaf_stream[q.w & ~PT_REF_LEAF];
                float hl[3], hh[3]; std::memcpy(hl, h, 12); std::memcpy(hh, h + 4, 12);
                const uint32_t first = refs[ch] & PT_LEAF_OFF_MASK, cnt = ((refs[ch] >> PT_LEAF_OFF_BITS) & (PT_LEAF_MAX_TRIS - 1u)) + 1u;
                bool ok = h[3] == first && h[7] == cnt;
                for (int k = 0; k < 3; k++) ok = ok && hl[k] == lo[ch][k] && hh[k] == hi[ch][k];
                for (uint32_t t = 0; t < cnt && ok; t++)
                    for (int j = 0; j < 3; j++) {
                        const float4 &v = b.tripos[3 * (size_t)(first + t) + j];
                        float g[3]; std::memcpy(g, h + 8 + 9 * t + 3 * j, 12);
                        ok = ok && std::memcmp(&g[0], &v.x, 4) == 0 && std::memcmp(&g[1], &v.y, 4) == 0 && std::memcmp(&g[2], &v.z, 4) == 0;
                    }
                if (!ok) bad_hdr++;
            } else if (q.w & PT_REF_LEAF) bad_hdr++;
            else {
                // an inner box is the exact union of its two children's boxes (what makes any topology equivalent, §3.2)
                if ((size_t)refs[ch] * 4 + 3 < b.fast_wnodes.size()) {
                    const float4 *cw = &b.fast_wnodes[(size_t)refs[ch] * 4];
                    const float clo[3] = {std::min(cw[0].x, cw[1].z), std::min(cw[0].y, cw[1].w), std::min(cw[0].z, cw[2].x)};
                    const float chi[3] = {std::max(cw[0].w, cw[2].y), std::max(cw[1].x, cw[2].z), std::max(cw[1].y, cw[2].w)};
                    for (int k = 0; k < 3; k++) if (clo[k] != lo[ch][k] || chi[k] != hi[ch][k]) { bad_hdr++; break; }
                }
                todo.push_back({refs[ch], q.w});
            }
        }
    }
    if (visited != b.qnodes.size() / 2) bad_hdr++;          // every node reached exactly once (a tree: no node can be reached twice)
    out[1] = (double)leaves; out[5] = viol; out[6] = boxes ? infl / (double)boxes : 0.0; out[7] = (double)bad_hdr;
    return PTMI_OK;
}

int ptmi_debug_build_image(const ptmi_triangle *tris, uint32_t nt, const ptmi_bvh_node *nodes, uint32_t nn, const ptmi_options *opt,
                           ptmi_image_info *info, float *wnodes16, uint32_t *qnodes8, float *tripos12, float *leafbox8) {
    if (!info || (nt && !tris) || (nn && !nodes)) return PTMI_E_INVALID;
    std::memset(info, 0, sizeof *info);
    ptmi_ctx tmp;                                   // host-only: never touches a device
    default_options(tmp.opt);
    if (opt) { tmp.opt.leaves = opt->leaves; tmp.opt.leaf_tris = opt->leaf_tris; tmp.opt.keep_reference_tree = opt->keep_reference_tree; }
    Built b;
    int rc = build_image(&tmp, tris, nt, nodes, nn, b);
    if (rc) { g_create_err = tmp.err; return rc; }
    const bool own = b.own, fast = own || !b.fast_wnodes.empty();
    const std::vector<float4> &walk = own ? b.own_tree.wnodes : fast ? b.fast_wnodes : b.wnodes;
    const std::vector<float4> &tp = own ? b.own_tree.tripos : b.tripos;
    const std::vector<uint4> &qn = own ? b.own_qnodes : b.qnodes;
    info->leaves_used = own ? 2u : 1u;
    info->n_wnodes = (uint32_t)(walk.size() / 4); info->n_tris = (uint32_t)(tp.size() / 3);
    info->root_ref = own ? b.own_tree.root_ref : fast ? b.fast_root : b.root_ref;
    info->depth = own ? b.own_tree.depth : fast ? b.fast_depth : b.depth;
    info->n_leaves = own ? b.own_tree.n_leaves : 0u;
    info->max_leaf_tris = own ? b.own_tree.max_leaf_tris : b.max_leaf_tris;
    info->quantised = qn.empty() ? 0u : 1u;
    for (int k = 0; k < 3; k++) {
        info->root_min[k] = own ? b.own_tree.root_min[k] : b.root_min[k]; info->root_max[k] = own ? b.own_tree.root_max[k] : b.root_max[k];
        info->q_origin[k] = b.q_origin[k]; info->q_scale[k] = b.q_scale[k];
    }
    info->pad = own ? b.own_tree.pad : 0.0f; info->safe_origin = own ? b.own_tree.safe_origin : 0.0f;
    info->ref_depth = b.depth;
    if (wnodes16 && !walk.empty()) std::memcpy(wnodes16, walk.data(), walk.size() * 16);
    if (qnodes8 && !qn.empty()) std::memcpy(qnodes8, qn.data(), qn.size() * 16);
    if (tripos12 && !tp.empty()) std::memcpy(tripos12, tp.data(), tp.size() * 16);
    if (leafbox8 && !b.leafbox.empty()) std::memcpy(leafbox8, b.leafbox.data(), b.leafbox.size() * 16);
    return PTMI_OK;
}

int ptmi_debug_math(ptmi_ctx *c, int op, uint32_t n, const float *a, const float *b, const float *cc, float *out) {
    if (!c || !a || !out) return PTMI_E_INVALID;
    if (n == 0) return PTMI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *da = nullptr, *db = nullptr, *dc = nullptr, *dout = nullptr;
    size_t bytes = (size_t)n * 4;
    HIP_TRY(c, hipMalloc(&da, bytes)); HIP_TRY(c, hipMalloc(&dout, bytes));
    HIP_TRY(c, hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    if (b) { HIP_TRY(c, hipMalloc(&db, bytes)); HIP_TRY(c, hipMemcpy(db, b, bytes, hipMemcpyHostToDevice)); }
    if (cc) { HIP_TRY(c, hipMalloc(&dc, bytes)); HIP_TRY(c, hipMemcpy(dc, cc, bytes, hipMemcpyHostToDevice)); }
    pt_launch_math(c->stream, op, n, da, db, dc, dout);
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    dfree(da); dfree(db); dfree(dc); dfree(dout);
    return PTMI_OK;
}

int ptmi_debug_exact_math(ptmi_ctx *c, int which, uint64_t *n_different, uint32_t *first_different) {
    if (!c || !n_different || which < 0 || which > 2) return PTMI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long *d = nullptr, h[2] = {0ull, ~0ull};
    HIP_TRY(c, hipMalloc(&d, sizeof h));
    HIP_TRY(c, hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    pt_launch_exact_math(c->stream, which, d);
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    dfree(d);
    *n_different = h[0];
    if (first_different) *first_different = (uint32_t)h[1];
    return PTMI_OK;
}

}  // extern "C"
