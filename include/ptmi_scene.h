/*
 * ptmi_scene.h — host-side scene preparation (C ABI, no GPU needed).
 *
 * The producers of the hot path's input blobs, restated from the reference's
 * TypeScript so that triangle order, node order and light order — which decide
 * tie-breaks and light picks per RNG seed — come out the same:
 *   ptmi_scene_sort_partially_*  <- src/utils/arr.ts:1-109  (sortArrayPartially)
 *   ptmi_scene_build_bvh         <- src/renderer/bvh.ts:53-229 + src/utils/aabb.ts
 *   ptmi_scene_emissive_lights   <- src/renderer/gpu.ts:121-138
 * Every function returns 0 on success; on failure a negative code, with the text
 * in ptmi_scene_last_error().
 */
#ifndef PTMI_SCENE_H
#define PTMI_SCENE_H

#include <stdint.h>
#include "ptmi_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

/* arr.ts:1-109 with compare = (a,b) => a-b (descending != 0: (a,b) => b-a).
 * Sorts arr[start, end) in place. Returns -1 ("Invalid indices") when
 * start < 0 || end > n || start >= end, like arr.ts:7-10 throws. */
int ptmi_scene_sort_partially_f64(double *arr, int64_t n, int64_t start, int64_t end, int descending);

/* Upper bound on the node count for n triangles (every leaf holds >= 1). */
uint32_t ptmi_scene_bvh_node_bound(uint32_t n_tris);

/* bvh.ts:53-157. Reorders tris in place (centroid sorts of bvh.ts:100-102) and
 * writes the nodes in the reference's order (root = 0, children appended
 * pairwise, right subtree built first). max_leaf = 4 and bins = 12 are the
 * reference defaults (bvh.ts:86, :110). max_depth_out (may be NULL) = depth of
 * the deepest leaf, root = 1. */
int ptmi_scene_build_bvh(ptmi_triangle *tris, uint32_t n_tris, uint32_t max_leaf, uint32_t bins,
                         ptmi_bvh_node *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes_out,
                         uint32_t *max_depth_out);

/* Threads ptmi_scene_build_bvh may use: 1 = the reference's single loop, 0 (default) = one per hardware thread, at
 * most 32. Subtrees over disjoint triangle ranges are built concurrently and spliced in the reference's node
 * order: the output is byte-identical for every thread count. */
void ptmi_scene_set_threads(int n_threads);

/* gpu.ts:121-138: one emissive light per triangle whose material has
 * length(emission) > 0, in ascending post-sort triangle index, appended after
 * lights_io[0 .. *n_lights_io). */
int ptmi_scene_emissive_lights(const ptmi_triangle *tris, uint32_t n_tris,
                               const ptmi_material *mats, uint32_t n_mats,
                               ptmi_light *lights_io, uint32_t lights_cap, uint32_t *n_lights_io);

const char *ptmi_scene_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
