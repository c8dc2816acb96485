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
