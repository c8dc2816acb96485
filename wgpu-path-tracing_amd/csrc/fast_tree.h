// fast_tree.h — host-side rebuild of the traversal hierarchy over the reference's leaves (see fast_tree.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

struct PtFastLeaf {
    float mn[3], mx[3];     // the leaf's own box, as stored in the reference node
    uint32_t ref;           // PT_REF_LEAF | (count-1) << 26 | first triangle
    uint32_t weight;        // triangle count (SAH weight)
};

// wnodes: 4 float4 per wide node in preorder (layout of pt_device.h); root_ref: wide-node index, or the
// leaf reference when there is a single leaf; depth: levels including the leaves.
void pt_build_fast_tree(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref,
                        uint32_t &depth);

// The same kind of hierarchy built on the device `s` belongs to (gpu_tree.hip: Morton-order linear BVH; ptmi_options.tree_builder = 2).
// Synchronises the stream. false: could not (allocation failure, non-finite centroids, more than 60 levels) — the caller builds on the host.
bool pt_build_fast_tree_gpu(const std::vector<PtFastLeaf> &leaves, std::vector<float4> &wnodes, uint32_t &root_ref, uint32_t &depth,
                            hipStream_t s);

// Quantised image of that hierarchy for the global traversal variant (layout: traverse.hip, QuantMem).
//   qnodes      2 uint4 per wide node: child boxes as 16-bit plane numbers on the grid origin + k * scale, rounded outward
//               (verified with the same fmaf the kernel evaluates), child references (leaf: PT_REF_LEAF | dword offset into
//               `stream`)
//   stream      per leaf: exact box (the reference node's), first triangle, count, then v0, e1, e2 of each triangle (9 dwords)
// tripos: 3 float4 per triangle (v0, e1, e2), indexed by triangle. Returns false when the hierarchy cannot be quantised
// (non-finite boxes, a stream beyond 2^31 dwords); the caller then keeps the exact image.
bool pt_quantize_tree(const std::vector<PtFastLeaf> &leaves, const std::vector<float4> &wnodes, const std::vector<float4> &tripos,
                      std::vector<uint4> &qnodes, std::vector<uint32_t> &stream, float origin[3], float scale[3],
                      uint32_t top_nodes, uint32_t &n_top);
// The quantised nodes are renumbered: the first n_top (<= top_nodes) are the top of the tree in breadth-first order (the
// kernel keeps them in LDS), the others follow in their preorder. The root stays node 0.

// ---- the library's OWN leaves (ptmi_options.leaves = 2) ------------------------------------------------------
// A full-sweep SAH hierarchy over the TRIANGLES themselves (bvh.ts:86-127 cuts leaves of <= 4 triangles from 11 equal-count candidates
// on one axis, and a ray then tests ~10 triangles where ~2.5 suffice), built down to single triangles and collapsed bottom-up into
// leaves of at most `max_leaf` triangles wherever the surface-area estimate says the leaf is cheaper than the box step.
//   wnodes   4 float4 per wide node, preorder; every child box PADDED outward by `pad`, so that the kernels' fused slab test
//            fma(bound, 1/d, -o/d) accepts every ray the exact box would (for origins within `safe_origin` of the coordinate origin)
//   tripos   3 float4 per listed triangle in LEAF order: (v0, bits(ORIGINAL triangle index)), (e1, 0), (e2, 0); a leaf reference
//            is PT_REF_LEAF | (count - 1) << 26 | position of its first triangle in this array
struct PtOwnTree {
    std::vector<float4> wnodes, tripos;
    uint32_t root_ref = 0xFFFFFFFFu, depth = 0, n_leaves = 0, max_leaf_tris = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};     // padded
    float pad = 0.0f, safe_origin = 0.0f;
};
struct ptmi_triangle;
// which: the original indices of the triangles to build over (those some reachable reference leaf lists), ascending.
// depth_limit: most levels (leaves included) the tree may have. false: a vertex is not finite (the caller keeps the reference's leaves).
bool pt_build_own_tree(const ptmi_triangle *tris, const std::vector<uint32_t> &which, uint32_t max_leaf, uint32_t depth_limit,
                       PtOwnTree &out);
// Quantised nodes of any wide-node hierarchy whose leaf references are to stay as they are (own leaves): 2 uint4 per node as in
// pt_quantize_tree, numbered with the top n_top <= top_nodes nodes first in breadth-first order, the rest in preorder.
bool pt_quantize_nodes(const std::vector<float4> &wnodes, std::vector<uint4> &qnodes, float origin[3], float scale[3],
                       uint32_t top_nodes, uint32_t &n_top);
