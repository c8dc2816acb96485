// gpu_tree.hip — the traversal hierarchy over the REFERENCE's leaves, built on the GPU (ptmi_options.tree_builder = 2).
//
// What is built is the same thing fast_tree.hip builds on the host (freedom 3): a binary hierarchy whose leaves are the
// reference's leaves (same triangle ranges, same exact boxes) and whose inner boxes are exact unions of them — any such hierarchy visits
// the same leaves and returns the same (t, triangle, u, v); only the number of box tests per ray depends on its quality. The reference's
// own builder (src/renderer/bvh.ts:53-157) cannot be parallelised bit for bit (its triangle order is the swap sequence of an unstable
// partial quicksort, src/utils/arr.ts); this is SURVEY.md §8(f1)'s "GPU LBVH" for the part of scene preparation that IS free.
//
// Algorithm: linear BVH. Morton code of each leaf's centroid (30 bits, on the centroid bounds), made unique by the leaf's index in the low
// 32 bits of a 64-bit key; radix sort of the keys (hipCUB); the binary radix tree over the sorted keys, every internal node found
// independently from the common-prefix lengths of its neighbours (Karras 2012); boxes fitted bottom-up, the second thread to arrive at a
// node (one atomic counter per node) unites its children. Output: the wide-node image of pt_device.h, renumbered in preorder on the host.
// Quality: a Morton-order tree tests more boxes per ray than the host's SAH tree — measured in profiles/README.md; the host builder
// stays the default.
#include "fast_tree.h"
#include "pt_device.h"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int TB = 256;

__device__ __forceinline__ uint32_t expand10(uint32_t v) {          // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void k_morton(uint32_t n, const PtFastLeaf *__restrict__ leaves, float3 lo, float3 inv_ext,
                         unsigned long long *__restrict__ keys) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const PtFastLeaf l = leaves[i];
    const float cx = (0.5f * l.mn[0] + 0.5f * l.mx[0] - lo.x) * inv_ext.x;
    const float cy = (0.5f * l.mn[1] + 0.5f * l.mx[1] - lo.y) * inv_ext.y;
    const float cz = (0.5f * l.mn[2] + 0.5f * l.mx[2] - lo.z) * inv_ext.z;
    auto q = [](float v) { v = v * 1024.0f; v = v < 0.0f ? 0.0f : (v > 1023.0f ? 1023.0f : v); return (uint32_t)v; };   // NaN -> 0
    const uint32_t code = (expand10(q(cx)) << 2) | (expand10(q(cy)) << 1) | expand10(q(cz));
    keys[i] = ((unsigned long long)code << 32) | i;
}

// length of the common prefix of keys i and j (keys are unique); -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012, one thread per internal node i of the n - 1: its range of sorted leaves, its split, its two children.
// child reference: internal -> node index; leaf -> PT_REF_LEAF | sorted position (replaced by the leaf's own reference in k_fit)
__global__ void k_radix_tree(int n, const unsigned long long *__restrict__ keys, uint32_t *__restrict__ child,
                             uint32_t *__restrict__ parent_int, uint32_t *__restrict__ parent_leaf) {
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t left = lo == gamma ? (PT_REF_LEAF | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t right = hi == gamma + 1 ? (PT_REF_LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    child[2 * i] = left; child[2 * i + 1] = right;
    if (left & PT_REF_LEAF) parent_leaf[gamma] = (uint32_t)i; else parent_int[gamma] = (uint32_t)i;
    if (right & PT_REF_LEAF) parent_leaf[gamma + 1] = (uint32_t)i; else parent_int[gamma + 1] = (uint32_t)i;
}

struct Box6 { float v[6]; };

// one thread per sorted leaf walks up; the second arrival at a node has both children's boxes behind it
__global__ void k_fit(int n, const unsigned long long *__restrict__ keys, const PtFastLeaf *__restrict__ leaves,
                      const uint32_t *__restrict__ child, const uint32_t *__restrict__ parent_int,
                      const uint32_t *__restrict__ parent_leaf, uint32_t *__restrict__ arrived, Box6 *__restrict__ node_box,
                      float4 *__restrict__ wnodes) {
    const int k = blockIdx.x * TB + threadIdx.x;
    if (k >= n) return;
    uint32_t node = parent_leaf[k];
    for (;;) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;            // the first to arrive leaves the node to the second
        __threadfence();
        Box6 b[2]; uint32_t ref[2];
        for (int c = 0; c < 2; c++) {
            const uint32_t r = child[2 * node + c];
            if (r & PT_REF_LEAF) {
                const PtFastLeaf l = leaves[(uint32_t)keys[r & ~PT_REF_LEAF]];          // low 32 bits of the key: the leaf's index
                b[c] = Box6{{l.mn[0], l.mn[1], l.mn[2], l.mx[0], l.mx[1], l.mx[2]}};
                ref[c] = l.ref;
            } else {
                const volatile float *q = node_box[r].v;            // written by the thread that finished that child, before its atomic
                for (int a = 0; a < 6; a++) b[c].v[a] = q[a];
                ref[c] = r;
            }
        }
        float4 *w = wnodes + 4 * (size_t)node;
        w[0] = make_float4(b[0].v[0], b[0].v[1], b[0].v[2], b[0].v[3]);
        w[1] = make_float4(b[0].v[4], b[0].v[5], b[1].v[0], b[1].v[1]);
        w[2] = make_float4(b[1].v[2], b[1].v[3], b[1].v[4], b[1].v[5]);
        w[3] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), 0.0f, 0.0f);
        Box6 u;
        for (int a = 0; a < 3; a++) { u.v[a] = fminf(b[0].v[a], b[1].v[a]); u.v[3 + a] = fmaxf(b[0].v[3 + a], b[1].v[3 + a]); }
        node_box[node] = u;
        if (node == 0u) return;                                     // the root of the radix tree
        node = parent_int[node];
    }
}

#define GT(expr) do { if ((expr) != hipSuccess) { ok = false; goto done; } } while (0)

}  // namespace

bool pt_build_fast_tree_gpu(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref, uint32_t &depth,
                            hipStream_t s) {
    const uint32_t n = (uint32_t)leaves.size();
    if (n < 2 || n > PT_LEAF_OFF_MASK) return false;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const PtFastLeaf &l : leaves)
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * l.mn[k] + 0.5f * l.mx[k];
            if (!std::isfinite(c)) return false;
            lo[k] = std::fmin(lo[k], c); hi[k] = std::fmax(hi[k], c);
        }
    float inv[3];
    for (int k = 0; k < 3; k++) { const float e = hi[k] - lo[k]; inv[k] = (e > 0.0f && std::isfinite(1.0f / e)) ? 1.0f / e : 0.0f; }

    bool ok = true;
    PtFastLeaf *d_leaves = nullptr; unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_child = nullptr, *d_pint = nullptr, *d_pleaf = nullptr, *d_arr = nullptr; Box6 *d_box = nullptr; float4 *d_w = nullptr;
    void *d_tmp = nullptr; size_t tmp_bytes = 0;
    const int blocks = (int)((n + TB - 1) / TB);
    std::vector<float4> raw;
    GT(hipMalloc(&d_leaves, (size_t)n * sizeof(PtFastLeaf)));
    GT(hipMalloc(&d_keys, (size_t)n * 8)); GT(hipMalloc(&d_keys2, (size_t)n * 8));
    GT(hipMalloc(&d_child, (size_t)(n - 1) * 8)); GT(hipMalloc(&d_pint, (size_t)n * 4)); GT(hipMalloc(&d_pleaf, (size_t)n * 4));
    GT(hipMalloc(&d_arr, (size_t)n * 4)); GT(hipMalloc(&d_box, (size_t)n * sizeof(Box6))); GT(hipMalloc(&d_w, (size_t)(n - 1) * 64));
    GT(hipMemcpyAsync(d_leaves, leaves.data(), (size_t)n * sizeof(PtFastLeaf), hipMemcpyHostToDevice, s));
    GT(hipMemsetAsync(d_arr, 0, (size_t)n * 4, s));
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(TB), 0, s, n, d_leaves, make_float3(lo[0], lo[1], lo[2]),
                       make_float3(inv[0], inv[1], inv[2]), d_keys);
    GT(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_keys2, (int)n, 0, 62, s));
    GT(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    GT(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_keys, d_keys2, (int)n, 0, 62, s));
    hipLaunchKernelGGL(k_radix_tree, dim3(blocks), dim3(TB), 0, s, (int)n, d_keys2, d_child, d_pint, d_pleaf);
    hipLaunchKernelGGL(k_fit, dim3(blocks), dim3(TB), 0, s, (int)n, d_keys2, d_leaves, d_child, d_pint, d_pleaf, d_arr, d_box, d_w);
    raw.resize((size_t)(n - 1) * 4);
    GT(hipMemcpyAsync(raw.data(), d_w, (size_t)(n - 1) * 64, hipMemcpyDeviceToHost, s));
    GT(hipStreamSynchronize(s));
    GT(hipGetLastError());
    {   // preorder renumbering (neighbours in the tree are neighbours in memory, as in the host builder's output) and the depth
        std::vector<uint32_t> renum(n - 1, 0xFFFFFFFFu), order; order.reserve(n - 1);
        std::vector<std::pair<uint32_t, uint32_t>> stack;          // (node, level)
        stack.emplace_back(0u, 1u);
        uint32_t dmax = 1;
        while (!stack.empty()) {
            const auto [nd, lv] = stack.back(); stack.pop_back();
            if (nd >= n - 1 || renum[nd] != 0xFFFFFFFFu) { ok = false; goto done; }           // not a tree: refuse
            renum[nd] = (uint32_t)order.size(); order.push_back(nd);
            uint32_t refs[2]; std::memcpy(&refs[0], &raw[(size_t)nd * 4 + 3].x, 4); std::memcpy(&refs[1], &raw[(size_t)nd * 4 + 3].y, 4);
            dmax = std::max(dmax, lv + 1u);                       // its children sit one level down (leaves included in the count)
            if (!(refs[1] & PT_REF_LEAF)) stack.emplace_back(refs[1], lv + 1u);
            if (!(refs[0] & PT_REF_LEAF)) stack.emplace_back(refs[0], lv + 1u);      // left on top: visited first
        }
        if (order.size() != n - 1 || dmax > 60u) { ok = false; goto done; }
        wnodes.assign((size_t)(n - 1) * 4, make_float4(0, 0, 0, 0));
        for (uint32_t k = 0; k < n - 1; k++) {
            const uint32_t nd = order[k];
            for (int q = 0; q < 3; q++) wnodes[(size_t)k * 4 + q] = raw[(size_t)nd * 4 + q];
            uint32_t refs[2]; std::memcpy(&refs[0], &raw[(size_t)nd * 4 + 3].x, 4); std::memcpy(&refs[1], &raw[(size_t)nd * 4 + 3].y, 4);
            for (uint32_t &r : refs) if (!(r & PT_REF_LEAF)) r = renum[r];
            float fl, fr; std::memcpy(&fl, &refs[0], 4); std::memcpy(&fr, &refs[1], 4);
            wnodes[(size_t)k * 4 + 3] = make_float4(fl, fr, 0.0f, 0.0f);
        }
        root_ref = 0u; depth = dmax;
    }
done:
    (void)hipFree(d_leaves); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_child); (void)hipFree(d_pint);
    (void)hipFree(d_pleaf); (void)hipFree(d_arr); (void)hipFree(d_box); (void)hipFree(d_w); (void)hipFree(d_tmp);
    if (!ok) (void)hipGetLastError();
    return ok;
}
