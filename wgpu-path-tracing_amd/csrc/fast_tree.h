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
