// pt_device.h — device-side data layout of the wavefront path tracer and the
// kernel launch interface shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ptmi_layout.h"

#define PT_HD __host__ __device__ inline

// ---- traversal image of the scene (built at upload from the 48-B reference nodes) ----
// Wide node, 64 B = 4 x float4: both children's boxes live in the parent, so one
// fetch decides both subtrees.
//   q0 = (Lmin.x, Lmin.y, Lmin.z, Lmax.x)
//   q1 = (Lmax.y, Lmax.z, Rmin.x, Rmin.y)
//   q2 = (Rmin.z, Rmax.x, Rmax.y, Rmax.z)
//   q3 = bits(Lref, Rref, 0, 0)
// Child reference: internal -> index of its wide node; leaf -> PT_REF_LEAF |
// (count-1) << 26 | first triangle. Triangle image: 3 x float4 (v0, e1 = v1-v0,
// e2 = v2-v0), the only triangle data the intersection test reads (pt.wgsl:128-156).
#define PT_REF_LEAF      0x80000000u
#define PT_REF_NONE      0xFFFFFFFFu
#define PT_LEAF_MAX_TRIS 32u
#define PT_LEAF_OFF_BITS 26u
#define PT_LEAF_OFF_MASK ((1u << PT_LEAF_OFF_BITS) - 1u)

struct DevScene;
struct DevScene {
    const ptmi_triangle *tris;  uint32_t n_tris;
    const ptmi_material *mats;  uint32_t n_mats;
    const ptmi_light *lights;   uint32_t n_lights;
    const void *atlas;          uint32_t atlas_w, atlas_h, atlas_fmt;   // 0 none, 1 rgba16f, 2 rgba32f
    const float4 *wnodes;       uint32_t n_wnodes;     // the hierarchy the kernels walk
    const float4 *tripos;
    float root_min[3], root_max[3];
    uint32_t root_ref;          // PT_REF_NONE: empty scene
    // When the reference tree is nested (every node box contains its children) the walked hierarchy is a
    // SAH tree rebuilt over the reference's leaves (fast_tree.hip, has_fast = 1) and the reference-shaped
    // image is kept for irregular rays; otherwise wnodes IS the reference-shaped image.
    const float4 *ref_wnodes;   uint32_t ref_root_ref, has_fast;
    // Quantised image of the rebuilt hierarchy for the global traversal variant (traverse.hip QuantMem; NULL: none):
    // 32-B nodes whose child boxes are 16-bit plane numbers on the grid origin + k * scale per axis, and the leaf stream
    // (per leaf: exact box + first triangle + count, then 9 dwords per triangle).
    const uint4 *qnodes;        const uint32_t *leaf_stream;
    float q_origin[3], q_scale[3];
    // |dx| + |dy| + |dz| of a ray up to which |e1 . (d x e2)| <= 2^100 for every triangle (2^98 / longest edge squared; 0 when an
    // edge is not finite): such rays take the triangle test whose short reciprocal has no range test (pt_math.h)
    float tri_safe_dsum;
    uint32_t q_cached;          // the first q_cached quantised nodes are the top levels in breadth-first order (kept in LDS)
    // OWN LEAVES (ptmi_options.leaves = 2, traverse_own.hip; own = 0: none of this is set). wnodes / tripos / qnodes / root_min / root_max
    // then describe the library's own hierarchy over the triangles: padded boxes, tripos in LEAF order with the original triangle
    // index in v0.w, n_own_tris entries; qnodes without a leaf stream. The tree as uploaded stays in ref_wnodes (its exact root box
    // in ref_root_min / _max) with the triangle images in ORIGINAL order in ref_tripos: what `slow` rays walk.
    uint32_t own, n_own_tris;
    const float4 *ref_tripos;
    const float4 *tri_leafbox;  // per ORIGINAL triangle index: (min.xyz, 0), (max.xyz, 0) of the reference leaf that lists it
    float ref_root_min[3], ref_root_max[3];
    float safe_origin;          // |o|_inf up to which the boxes' padding covers the rounding of the fused slab test
    // compact references (scenes of up to 4 096 triangles, NULL otherwise): copies of wnodes / ref_wnodes whose child references are 16 bits —
    // an internal node's index, or 0x8000 | (count - 1) << 12 | first triangle — for the kernels with 16-bit stack entries
    const float4 *wnodes16, *ref_wnodes16;
    uint32_t root_ref16, ref_root_ref16;
    const uint4 *qnodes16;      // qnodes with the same 16-bit references (NULL: none)
    unsigned long long *verify_stat;    // += rays whose winner failed its reference leaf's box and were traced again
    const DevScene *self;       // this description in device memory (the own-leaf kernels read it from there, not from kernel arguments)
};

// ---- path state: 56 B per path, four streams indexed by path id ----
//   O = (origin.xyz, bits(rng state))   D = (direction.xyz, throughput.x)      the two float4 `extend` reads
//   C = (throughput.y, throughput.z)    L = (radiance.xyz, 0)
// Radiance L and the contribution SC of a shadow record have three lanes and are stored as three floats.
// SC (read and written in queue-slot order) sits at 12-byte stride. L is read-modify-written by path id, scattered, and its
// stride is chosen per dispatch (DevPaths::l_stride, in floats): 3 where the scene lives in LDS, 4 where the traversal
// kernels walk it from memory and every extra line the RMW straddles competes with node fetches. Measured, interleaved runs,
// Msamples/s (both 16 B / both 12 B / L 16 + SC 12 / L 12 + SC 16): config 1 8 907 / 9 172 / 9 031 / 9 133; config 3
// 4 742 / 4 524 / 4 734 / 4 482.
struct rgb_sc { float x, y, z; };
struct DevPaths {
    float4 *O, *D; float2 *C; float *L; uint32_t l_stride = 3;
    // stride 4 means whole 16-byte accesses (a 12-byte access is issued as two requests: with stride 4 and 12-byte accesses
    // raygen's store of L takes 1.9 instead of 1.1 ms per 64 spp, and config 3 loses the same 5 % as with stride 3)
    __device__ __forceinline__ rgb_sc ldL(uint32_t p) const {
        if (l_stride == 4u) { const float4 v = reinterpret_cast<const float4 *>(L)[p]; return rgb_sc{v.x, v.y, v.z}; }
        return reinterpret_cast<const rgb_sc *>(L)[p];
    }
    __device__ __forceinline__ void stL(uint32_t p, float x, float y, float z) const {
        if (l_stride == 4u) reinterpret_cast<float4 *>(L)[p] = make_float4(x, y, z, 0.0f);
        else reinterpret_cast<rgb_sc *>(L)[p] = rgb_sc{x, y, z};
    }
};
// hit record, 8 B per queue slot: (t, bits(triangle index)); t = -1 on a miss. `shade` rebuilds (u, v) from the triangle.
// shadow record, 44 B per queue slot:
//   SO = (origin.xyz, dist or -1 for directional)  SD = (wi.xyz, bits(path id))
//   SC = throughput * directLight .xyz (12-byte stride)   added to L[path] when unoccluded
// One allocation per bounce parity: SO at the base, SD `cap` float4 further, SC after both (the shadow kernel carries only the
// base and cap: two pointers fewer in scalar registers, see traverse.hip ShadowIO)
struct DevShadow { float4 *SO, *SD; rgb_sc *SC; uint32_t cap; };   // 44 B per queue slot

// The rows one context renders: [y0, y1) of a width x height frame, or — when parts > 1 — every parts-th strip
// of `strip` rows inside that range, starting with strip number `part` (row bands interleaved across GPUs so each
// sees a sample of the whole picture). rows = how many rows that is; local row l is frame row row_of(l).
struct DevBand {
    uint32_t width, height, y0, y1, strip, parts, part, rows;
    PT_HD uint32_t row_of(uint32_t l) const {
        if (parts <= 1u) return y0 + l;
        return y0 + ((l / strip) * parts + part) * strip + l % strip;
    }
};

struct ShadeParams {
    uint32_t bounce, max_bounces, do_mis;
    unsigned long long *stats;          // [1] += next-event samples counted but not traced (zero contribution)
    uint32_t emit_records;              // 1: an emissive hit does not add to L here; it leaves a record (SO.w = -2: nothing to trace)
                                        //    that `shadow` adds like an unoccluded light sample — all additions to L then happen in
                                        //    that one kernel, in bounce order, and `shadow` can run beside the next bounce's kernels.
                                        //    stats[3] += such records (they are not shadow rays)
};

enum { PT_VARIANT_GLOBAL = 1, PT_VARIANT_LDS = 2, PT_VARIANT_LDS_NODES = 3,
       // own leaves (traverse_own.hip): exact / quantised nodes in LDS, with (…_LDS) or without (…_NODES) the triangle images, or from memory
       PT_VARIANT_OWN_LDS = 4, PT_VARIANT_OWN_LDS_NODES = 5, PT_VARIANT_OWN_QLDS = 6, PT_VARIANT_OWN_QLDS_NODES = 7,
       PT_VARIANT_OWN_QGLOBAL = 8, PT_VARIANT_OWN_GLOBAL = 9,
       PT_VARIANT_OWN_LDS16_NODES = 10,         // exact nodes with 16-bit references and 16-bit stack entries (scenes up to 4 096 triangles)
       PT_VARIANT_OWN_QLDS16_NODES = 11 };      // quantised nodes with 16-bit references: two workgroups per CU for trees of up to 2 046 nodes,
                                                // 8 - 15 16-bit entries per lane (what the nodes leave of 80 KB), the node stack spills

struct TraverseConfig {
    int variant;            // PT_VARIANT_*
    int stack_entries;      // 15 (two workgroups per CU only), 16, 32 or 64; PT_VARIANT_OWN_QLDS16_NODES: 8 ... 15
    int cull;               // 0/1
    size_t lds_scene_bytes; // LDS variant: bytes of wnodes + tripos
    int wgs_per_cu;         // node cache: 2 (small trees, whole stack in LDS) or 1 (mid-size trees, spilling stacks)
    uint32_t *spill;        // global variant: per-lane overflow of the node stack, pt_spill_bytes(blocks) bytes
    int wants_spill;        // the variant needs one (the caller supplies `spill`: each concurrently running kernel its own)
    int quantized;          // global variant: walk the quantised image when the scene has one
};
#ifndef PT_QCACHE_NODES
#define PT_QCACHE_NODES 256      /* quantised nodes of the top levels staged in LDS per workgroup (8 KB) */
#endif
#define PT_SPILL_ENTRIES 64     /* >= the deepest node stack: upload rejects trees deeper than 62 */
size_t pt_spill_bytes(int blocks);

// ---- launchers (each enqueues on `s`; grids are persistent, sized by the caller) ----
void pt_launch_raygen(hipStream_t s, int blocks, const ptmi_camera &cam, DevBand band, uint32_t frame0,
                      uint32_t n_frames, DevPaths p, uint32_t *count_out);
void pt_launch_raygen_list(hipStream_t s, const ptmi_camera &cam, uint32_t n, const uint32_t *xs,
                           const uint32_t *ys, const uint32_t *frames, DevPaths p);
void pt_launch_extend(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      const uint32_t *queue, const uint32_t *count, float2 *hits);
void pt_launch_extend_own(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                          const uint32_t *queue, const uint32_t *count, float2 *hits);        // traverse_own.hip
// (u, v) of n hit records, rebuilt the way `shade` does it (debug entry point of the parity tests)
void pt_launch_hit_uv(hipStream_t s, uint32_t n, const DevScene &sc, DevPaths p, const float2 *hits, float2 *uv);
// shadow_queue: slots of the shadow records to trace (NULL = slots 0..count-1), count = their number
void pt_launch_shadow(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                      DevShadow sh, const uint32_t *shadow_queue, const uint32_t *count, uint8_t *occluded_out);
void pt_launch_shadow_own(hipStream_t s, int blocks, const TraverseConfig &cfg, const DevScene &sc, DevPaths p,
                          DevShadow sh, const uint32_t *shadow_queue, const uint32_t *count, uint8_t *occluded_out);
void pt_launch_shade(hipStream_t s, int blocks, const DevScene &sc, DevPaths p, const uint32_t *queue,
                     const uint32_t *count, const float2 *hits, DevShadow sh, uint64_t *alive_mask,
                     uint64_t *shadow_mask, ShadeParams sp);
void pt_launch_shade_fast(hipStream_t s, int blocks, const DevScene &sc, DevPaths p, const uint32_t *queue,
                          const uint32_t *count, const float2 *hits, DevShadow sh, uint64_t *alive_mask,
                          uint64_t *shadow_mask, ShadeParams sp);      // perf mode: shade.hip built with fast division / sqrt
// ordered stream compaction of the survivors: masks -> next queue + its count, plus statistics
// (tiles = ceil(capacity / pt_compact_tile_slots()) + 1: one workgroup per tile of ballot words)
uint32_t pt_compact_tile_slots(void);
void pt_launch_compact(hipStream_t s, int tiles, const uint32_t *queue, const uint32_t *count,
                       const uint64_t *alive_mask, const uint64_t *shadow_mask, uint32_t *tile_sums,
                       uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_queue, uint32_t *shadow_count,
                       unsigned long long *stats, uint32_t bounce, int do_scatter);
void pt_launch_accumulate(hipStream_t s, int blocks, DevBand band, uint32_t frame0, uint32_t n_frames,
                          const float *L, uint32_t l_stride, float4 *out);
void pt_launch_blit(hipStream_t s, int blocks, uint32_t W, uint32_t H, const float4 *color, float4 *out_f32,
                    uint32_t *out_rgba8);
// a device's rows of the frame <-> a contiguous buffer (ptmi_multi_gather)
void pt_launch_pack_rows(hipStream_t s, int blocks, DevBand band, const float4 *frame, float4 *packed);
void pt_launch_unpack_rows(hipStream_t s, int blocks, DevBand band, const float4 *packed, float4 *frame);
// the rows DevBand describes for a context with these options on a width x height frame (rows = 0: none)
struct ptmi_options;
DevBand pt_band_of(const ptmi_options &opt, uint32_t width, uint32_t height);
// what ptmi_multi.hip needs from a context (ptmi_api.hip)
struct ptmi_ctx;
struct PtPrepared;               // a scene prepared on the host: validation + traversal image (ptmi_api.hip)
PtPrepared *pt_prepare_scene(ptmi_ctx *c, const ptmi_triangle *tris, uint32_t nt, const ptmi_material *mats, uint32_t nm,
                             const ptmi_bvh_node *nodes, uint32_t nn, const ptmi_light *lights, uint32_t nl, int *rc_out);
int pt_install_scene(ptmi_ctx *c, const PtPrepared *p);      // allocates and copies on c's device; c keeps its old scene on failure
void pt_free_prepared(PtPrepared *p);
hipStream_t pt_ctx_stream(ptmi_ctx *c);
float4 *pt_ctx_output(ptmi_ctx *c);
int pt_ctx_device(const ptmi_ctx *c);
int pt_ctx_cus(const ptmi_ctx *c);
void pt_launch_exact_math(hipStream_t s, int which, unsigned long long *out);
void pt_launch_math(hipStream_t s, int op, uint32_t n, const float *a, const float *b, const float *c, float *out);

