/*
 * ptmi.h — C ABI of the MI355X wavefront path tracer (libptmi.so).
 *
 * Drop-in boundary for the reference's compute pass: the bind group of
 * src/renderer/renderer.ts:368-381 (bindings 0..6 of src/shader/pt.wgsl:104-110)
 * plus the per-frame camera write and dispatch of renderer.ts:403-431. Every
 * blob is the exact WGSL-layout byte image the reference uploads
 * (include/ptmi_layout.h). Plain pointers and sizes only — no HIP or torch
 * types; a caller binds this with cgo / JNI / N-API / ctypes.
 *
 * There is no CPU backend behind this ABI. ptmi_create fails when no gfx950
 * device is present.
 *
 * All functions return 0 on success and a negative PTMI_E_* code on failure;
 * ptmi_last_error(ctx) (ctx may be NULL for creation errors) gives the text.
 * A context is not thread-safe (like the reference's single-queue Renderer);
 * work is ordered on one HIP stream per context.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>
#include "ptmi_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_ABI_VERSION 4

enum {
    PTMI_OK = 0,
    PTMI_E_INVALID = -1,     /* bad argument / malformed blob */
    PTMI_E_NODEVICE = -2,    /* no usable GPU */
    PTMI_E_HIP = -3,         /* a HIP runtime call failed */
    PTMI_E_STATE = -4,       /* call order (e.g. dispatch before upload/resize) */
    PTMI_E_UNSUPPORTED = -5  /* scene exceeds an implementation limit */
};

enum { PTMI_ATLAS_RGBA16F = 1, PTMI_ATLAS_RGBA32F = 2 };   /* reference uploads rgba16float (renderer.ts:246-261) */
/* AUTO: the scene's traversal image lives in LDS when it fits, else it is walked from global memory. GLOBAL / LDS force
 * one (LDS fails for scenes that do not fit). GLOBAL_EXACT: global memory, on the exact 64-byte nodes instead of the
 * quantised 32-byte image GLOBAL uses (same results; kept for tests and comparisons). */
enum { PTMI_TRAVERSAL_AUTO = 0, PTMI_TRAVERSAL_GLOBAL = 1, PTMI_TRAVERSAL_LDS = 2, PTMI_TRAVERSAL_GLOBAL_EXACT = 3 };

typedef struct ptmi_ctx ptmi_ctx;

/* Runtime options. The reference fixes these at shader-compile time
 * (pt.wgsl:5 MAX_BOUNCES = 8, :636 DO_MIS = true). */
typedef struct ptmi_options {
    uint32_t max_bounces;       /* 1..64; default 8 */
    uint32_t do_mis;            /* 0/1; default 1 */
    uint32_t tile_y0, tile_y1;  /* rows [y0,y1) this context renders; y1 = 0 -> height. Other rows are untouched */
    uint32_t frames_per_batch;  /* frames traced together as one wavefront batch; 0 -> auto (up to 64 frames / ~128 Mi paths, ~23 GB of
                                   path state at 1920x1080; a dispatch of fewer frames is one smaller batch) */
    uint32_t traversal;         /* PTMI_TRAVERSAL_*; AUTO picks LDS when the scene fits */
    uint32_t cull;              /* 1 (default): ordered traversal with conservative distance cull;
                                   0: every box-overlapping leaf is tested, as pt.wgsl:248-291 does */
    uint32_t timing;            /* 0: none; 1: HIP events around each dispatch (gpu_ms); 2: also around every
                                   extend launch (extend_ms); 3: around every kernel launch (extend, shade, shadow,
                                   raygen, compaction, accumulate — the per-kernel times then add up to gpu_ms).
                                   Each event pair costs a few microseconds of stream time. */
    uint32_t keep_reference_tree; /* read by ptmi_upload_scene. 0 (default): when every node box of the uploaded BVH contains
                                   its children's boxes, traversal walks a SAH hierarchy rebuilt over the SAME leaves (same
                                   results, about half the box tests); 1: always walk the tree as uploaded */
    uint32_t tile_parts, tile_part, tile_strip; /* interleaved row sharding: with tile_parts = N > 1 this context renders, of the
                                   rows [tile_y0, tile_y1), the strips of tile_strip rows (0 -> 1) numbered tile_part,
                                   tile_part + N, ... — N contexts with tile_part = 0..N-1 cover the range exactly once, each
                                   with an even sample of the picture. 0 / 1 = all rows of the range (default) */
    uint32_t perf_mode;         /* 0 (default): the parity arithmetic of DESIGN.md §3 — results equal the oracle's bit for bit.
                                   1: `shade` may use the device's fast reciprocal / reciprocal square root / square root
                                   (1 ulp) instead of the correctly rounded forms. Same RNG streams and control flow; radiance
                                   agrees statistically (tests/test_gpu_perf_mode.py), not bit for bit. Never the headline. */
    uint32_t reserved_a;        /* must be 0 (ABI <= 3: ray_sort — survivors grouped by direction octant; measured slower and removed,
                                   profiles/README.md "Removed in round 4") */
    uint32_t overlap;           /* 0: every kernel of a dispatch on the context's one stream, in order; 1: the any-hit `shadow` kernel of
                                   bounce b runs on a second stream beside `extend` / `shade` of bounce b + 1 — it is
                                   the only kernel that adds to the radiance then, in bounce order, so results are unchanged;
                                   2 = library default (currently 1) */
    uint32_t reserved_b[4];     /* must be 0 (ABI 3: worklist, tails, state, pipeline — four ways to compute the same bits, each measured
                                   slower or level and removed in round 4; numbers and the last commit that had them: profiles/README.md) */
    uint32_t tree_builder;      /* read by ptmi_upload_scene: who builds the traversal hierarchy over the uploaded leaves (when
                                   keep_reference_tree = 0). 1 = the host (full-sweep / binned SAH + rotations, threaded: 137 ms for the
                                   334 174 leaves of the 1 M-triangle scene); 2 = the GPU (Morton-order linear BVH: radix sort + radix
                                   tree + bottom-up fit, a few ms) — the same leaves and exact unions, hence the same results; a
                                   Morton tree tests more boxes per ray (profiles/README.md). 0 = library default (1) */
    /* ABI 4 */
    uint32_t leaves;            /* read by ptmi_upload_scene: which leaves the traversal kernels test triangles in.
                                   1 = the uploaded BVH's own leaves (bvh.ts:86-127 cuts <= 4 triangles from 11 equal-count candidates on one
                                       axis): a triangle is tested iff the box of ITS reference leaf passes, exactly as pt.wgsl:248-291 does;
                                   2 = the library's own leaves: a full-sweep SAH hierarchy over the TRIANGLES (leaf_tris at most per leaf),
                                       boxes padded outward so that they are conservative for the kernels' fused slab test. A ray then tests
                                       a quarter of the triangles (Cornell: 9.8 -> 2.4 per closest-hit ray). The closest hit is verified
                                       against the reference leaf's box before it is reported, and a ray whose winner fails that test — or
                                       whose direction has a zero / non-finite component, or whose origin lies far outside the scene — is
                                       traced again over the uploaded tree, so results equal mode 1's;
                                   0 = library default (2) */
    uint32_t leaf_tris;         /* leaves = 2: most triangles per own leaf, 1 .. 32; 0 = library default (measured: profiles/README.md) */
    uint32_t reserved[1];       /* must be 0 */
} ptmi_options;

typedef struct ptmi_stats {
    uint64_t paths;             /* (pixel, frame) samples traced since the last reset */
    uint64_t segments;          /* path segments = bounce-loop iterations reaching sceneIntersect (pt.wgsl:643-644) */
    uint64_t shadow_rays;       /* shadow traversals of the reference (pt.wgsl:392/421/463). Those whose contribution is
                                   exactly zero (light behind the surface) are counted here but not traced: they cannot
                                   change the radiance */
    uint64_t dispatches;        /* ptmi_dispatch calls */
    uint64_t frames;            /* frames traced */
    uint64_t segments_by_bounce[64];
    double   gpu_ms;            /* device time of all dispatches (HIP events on the context's stream) */
    double   extend_ms;         /* ... of the closest-hit traversal kernel only, and its launch count */
    uint64_t extend_launches;
    double   shade_ms;
    double   shadow_ms;
    uint32_t bvh_depth;         /* of the uploaded tree */
    uint32_t traversal_used;    /* PTMI_TRAVERSAL_GLOBAL or _LDS */
    uint32_t frames_per_batch_used;
    uint32_t radiance_stride_bytes;  /* of the last dispatch's per-path radiance buffer: 12, or 16 for scenes walked from memory (was reserved) */
    /* ABI 2 */
    uint64_t shadow_traced;     /* shadow records the any-hit kernel actually traced (shadow_rays minus the zero-contribution ones) */
    uint64_t shade_launches, shadow_launches;
    double   raygen_ms, compact_ms, accumulate_ms;   /* timing >= 3 */
    double   upload_ms;         /* wall time of the last ptmi_upload_scene, and its parts: validation + traversal image, */
    double   upload_tree_ms;    /* ... the hierarchy rebuilt over the uploaded leaves, */
    double   upload_copy_ms;    /* ... host-to-device copies */
    /* ABI 4 (ABI 3 had worklist_used, tails_used, state_used, pipeline_used here) */
    uint32_t leaves_used;       /* 1 / 2: ptmi_options.leaves as the uploaded scene's traversal image was built */
    uint32_t leaf_tris_used;    /* most triangles in a leaf of that image */
    uint32_t extend_variant, shadow_variant;   /* of the last dispatch: PTMI_VARIANT_* the closest-hit / any-hit kernel ran as */
    /* leaves = 2: closest hits / occluders whose reference leaf's box did not pass and rays that were therefore traced again over the
     * uploaded tree (both kernels together), since the last reset */
    uint64_t verify_failed;
    uint32_t reserved_stats[2];
} ptmi_stats;

/* ---- lifetime ----------------------------------------------------------- */
int ptmi_abi_version(void);
/* device_ordinal >= 0 selects the HIP device. Replaces requestAdapter/requestDevice +
 * createPipelines (renderer.ts:203-214, :513-533). */
int ptmi_create(int device_ordinal, ptmi_ctx **out);
int ptmi_destroy(ptmi_ctx *ctx);                              /* renderer.ts:482-494 destroy() */
const char *ptmi_last_error(const ptmi_ctx *ctx);

/* ---- resources (createBuffers, renderer.ts:242-355) --------------------- */
/* The four storage buffers of bindings 1, 2, 4, 5. Host blobs are copied during the call. */
int ptmi_upload_scene(ptmi_ctx *ctx,
                      const ptmi_triangle *triangles, uint32_t n_triangles,
                      const ptmi_material *materials, uint32_t n_materials,
                      const ptmi_bvh_node *bvh_nodes, uint32_t n_nodes,
                      const ptmi_light *lights, uint32_t n_lights);
/* Binding 6. texels: width*height RGBA, row-major, texel (0,0) first. NULL/0 removes the atlas. */
int ptmi_upload_atlas(ptmi_ctx *ctx, const void *texels, uint32_t width, uint32_t height, int format);
/* Binding 0: (re)allocates the width*height*16-byte output buffer, zero-filled (renderer.ts:272-279, :496-510). */
int ptmi_resize(ptmi_ctx *ctx, uint32_t width, uint32_t height);
int ptmi_set_options(ptmi_ctx *ctx, const ptmi_options *opt);
int ptmi_get_options(const ptmi_ctx *ctx, ptmi_options *opt);

/* ---- the compute pass (updateCamera + dispatch, renderer.ts:403-431) ---- */
/* Traces n_frames consecutive frames starting at camera->frame_index (one sample per
 * pixel per frame, pt.wgsl:719) and folds them into the output buffer in frame
 * order (pt.wgsl:753-761). Equals n_frames single-frame dispatches with the
 * frame index incremented by the caller (renderer.ts:453). Asynchronous. */
int ptmi_dispatch(ptmi_ctx *ctx, const ptmi_camera *camera, uint32_t n_frames);
int ptmi_synchronize(ptmi_ctx *ctx);
/* Back-pressure for a caller that enqueues dispatches from a loop without reading results — a preview loop; the reference paces
 * itself on requestAnimationFrame (renderer.ts:456-473). Blocks until at most max_in_flight of this context's dispatches have
 * not yet finished on the device and reports how many still are (in_flight may be NULL). max_in_flight = 0xFFFFFFFF only polls.
 * Independently of this call the library never lets more than 256 dispatches queue up: ptmi_dispatch then waits for the oldest. */
int ptmi_throttle(ptmi_ctx *ctx, uint32_t max_in_flight, uint32_t *in_flight);

/* ---- output buffer (binding 0) ------------------------------------------ */
/* width*height float4 (xyz = running mean, w = 0), index y*width+x. Synchronises. */
int ptmi_read_output(ptmi_ctx *ctx, float *dst_rgba, size_t n_floats);
int ptmi_write_output(ptmi_ctx *ctx, const float *src_rgba, size_t n_floats);
/* Device-side access for zero-copy consumers (blit, RCCL gather): the buffer's
 * device address, or a caller-owned device buffer of >= width*height*16 bytes to
 * render into instead (like binding a GPUBuffer the caller created). */
void *ptmi_output_device_ptr(ptmi_ctx *ctx);
int ptmi_bind_output_device(ptmi_ctx *ctx, void *device_ptr, size_t bytes);
/* Order this context's work on a caller-owned hipStream_t (NULL = the context's own). */
int ptmi_set_stream(ptmi_ctx *ctx, void *hip_stream);

/* ---- presentation (the reference's blit pass, src/shader/blit.wgsl:43-155; renderer.ts:434-449) ---- */
/* Tone-maps the output buffer (exposure 2^1, AgX, gamma 1/2.2) into a width*height canvas, row 0 = top.
 * dst_rgba_f32 (n_floats must be width*height*4, alpha 1) and/or dst_rgba8 (n_bytes must be width*height*4);
 * either pointer may be NULL (its count is then ignored). A wrong count is PTMI_E_INVALID: nothing is written.
 * Synchronises. Uses the device's log2/pow: compared with a tolerance, not bit for bit. */
int ptmi_blit(ptmi_ctx *ctx, float *dst_rgba_f32, size_t n_floats, uint8_t *dst_rgba8, size_t n_bytes);
/* Size of the output buffer as last set by ptmi_resize (0 x 0 before). */
int ptmi_get_size(const ptmi_ctx *ctx, uint32_t *width, uint32_t *height);

/* ---- several GPUs of one node behind the same boundary (BASELINE configs[4]) ----------------------------------
 * The caller this serves is the frame loop of src/renderer/renderer.ts:415-454: one host thread, one `Renderer`, now over N
 * devices. Pixels are independent (pt.wgsl:753-761) and the RNG is seeded per (x, y, frame) (random.wgsl:3-5), so the frame's
 * rows are dealt to the devices as interleaved strips (tile_parts / tile_part / tile_strip above) with no data-path collective:
 * every device holds the whole scene, traces all frames of ITS rows and accumulates them locally. ptmi_multi_gather packs each
 * device's rows into one contiguous buffer, moves them to device 0 with ONE RCCL gather over xGMI (ncclGather,
 * /opt/rocm/include/rccl/rccl.h:745; single process, ncclCommInitAll over the listed devices) and unpacks them by row index into
 * device 0's output buffer, which then holds the frame exactly as one device would have rendered it — bit for bit.
 * librccl.so.1 is loaded when the first multi-device handle is created (not by single-device users of this library).
 * All calls are made from one host thread; work is enqueued asynchronously on one stream per device (ptmi_multi_dispatch itself
 * enqueues every device from a thread of its own for the duration of the call: ~0.3 ms of host time per device and 64-frame batch).
 * STATUS: what a one-GPU machine can check is checked — N = 1 through RCCL gives the un-sharded bits; N = 2 .. 8 contexts on one
 * device with copies in place of the collective (PTMI_MULTI_LOOPBACK) assemble the single-device frame bit for bit. The real N > 1
 * path (ncclCommInitAll over several devices, one grouped ncclGather with a NULL receive buffer on the non-roots, the unpack of slots
 * 1 .. N-1) HAS NEVER RUN: every machine this library was built and tested on has one GPU. tests/test_gpu_multi.py holds the
 * 2-device test; it skips itself below two GPUs. */
typedef struct ptmi_multi ptmi_multi;
enum {
    PTMI_MULTI_LOOPBACK = 1u    /* device-to-device copies in place of the collective: lets ONE device stand in for several (the same
                                   ordinal may then be listed more than once) — for tests of the packing on a one-GPU box; RCCL is not loaded */
};
/* ordinals: n_devices HIP device ordinals (NULL = 0 .. n_devices-1); the first one is the root that ends up with the frame. */
int ptmi_multi_create(int n_devices, const int *ordinals, uint32_t flags, ptmi_multi **out);
int ptmi_multi_destroy(ptmi_multi *m);
const char *ptmi_multi_last_error(const ptmi_multi *m);       /* m may be NULL for creation errors */
int ptmi_multi_count(const ptmi_multi *m);
ptmi_ctx *ptmi_multi_context(ptmi_multi *m, int i);           /* device i's context (statistics, per-stage entry points); owned by m */
/* ptmi_upload_scene / _atlas / ptmi_resize on every device (the scene is replicated: <= 163 MB in BASELINE's configs) */
int ptmi_multi_upload_scene(ptmi_multi *m,
                            const ptmi_triangle *triangles, uint32_t n_triangles,
                            const ptmi_material *materials, uint32_t n_materials,
                            const ptmi_bvh_node *bvh_nodes, uint32_t n_nodes,
                            const ptmi_light *lights, uint32_t n_lights);
int ptmi_multi_upload_atlas(ptmi_multi *m, const void *texels, uint32_t width, uint32_t height, int format);
int ptmi_multi_resize(ptmi_multi *m, uint32_t width, uint32_t height);
/* Options for every device. tile_parts / tile_part are set by the library (device i renders the strips i, i + N, ...);
 * tile_strip = 0 picks the strip height: 4 rows, or the largest smaller height that makes the frame a whole number of rounds
 * (3840x2160 over 8 devices: 3), so that all devices get equal shares; tile_y0 / tile_y1 must be 0. */
int ptmi_multi_set_options(ptmi_multi *m, const ptmi_options *opt);
int ptmi_multi_get_options(const ptmi_multi *m, ptmi_options *opt);
/* ptmi_dispatch on every device, each for its own strips. Asynchronous. */
int ptmi_multi_dispatch(ptmi_multi *m, const ptmi_camera *camera, uint32_t n_frames);
/* Assembles the frame in device 0's output buffer (pack -> ncclGather -> unpack), ordered after the dispatches so far on every
 * device's stream. Asynchronous. Call it after the last frame, or every k frames for a preview (the rows keep accumulating on
 * their devices; the gather only copies). With one device it is a no-op. */
int ptmi_multi_gather(ptmi_multi *m);
int ptmi_multi_synchronize(ptmi_multi *m);
int ptmi_multi_throttle(ptmi_multi *m, uint32_t max_in_flight, uint32_t *in_flight);   /* ptmi_throttle on every device; reports the maximum */
/* gather + synchronise + copy device 0's output buffer out (width*height float4) */
int ptmi_multi_read_output(ptmi_multi *m, float *dst_rgba, size_t n_floats);
/* resume: the frame is written to every device (each keeps accumulating its rows on top of it) */
int ptmi_multi_write_output(ptmi_multi *m, const float *src_rgba, size_t n_floats);
/* gather + the blit pass on device 0 (see ptmi_blit) */
int ptmi_multi_blit(ptmi_multi *m, float *dst_rgba_f32, size_t n_floats, uint8_t *dst_rgba8, size_t n_bytes);
/* counters summed over the devices; times (gpu_ms, *_ms) are the maximum over the devices; the rest is device 0's */
int ptmi_multi_get_stats(ptmi_multi *m, ptmi_stats *out);
int ptmi_multi_reset_stats(ptmi_multi *m);
/* time of the last ptmi_multi_gather on device 0's stream — from the moment every device has rendered its rows: pack + collective +
 * unpack, not the wait for the slowest device — in ms (-1 before the first; synchronises) */
int ptmi_multi_gather_ms(ptmi_multi *m, double *ms);

/* ---- statistics ----------------------------------------------------------- */
int ptmi_get_stats(ptmi_ctx *ctx, ptmi_stats *out);           /* synchronises */
int ptmi_reset_stats(ptmi_ctx *ctx);

/* ---- per-stage entry points (parity tests of single kernels) -------------- */
/* raygen kernel (pt.wgsl:714-750) for explicit (x, y, frame) triples. o3/d3: n*3 floats. */
int ptmi_debug_raygen(ptmi_ctx *ctx, const ptmi_camera *camera, uint32_t n, const uint32_t *xs,
                      const uint32_t *ys, const uint32_t *frames, float *o3, float *d3, uint32_t *rng);
/* extend kernel (pt.wgsl:248-296) on caller rays: t = -1 / tri = 0xFFFFFFFF on miss. */
int ptmi_debug_intersect(ptmi_ctx *ctx, uint32_t n, const float *o3, const float *d3,
                         float *t, uint32_t *tri, float *u, float *v);
/* shadow kernel predicate (pt.wgsl:394/423/465); dist[i] < 0 = directional light (any hit occludes). Every negative value
 * means that: the library normalises them to -1 before the kernel sees them (inside a dispatch, -2 marks the record of an
 * emissive hit, which is added without a traversal — never a value a caller can inject here). */
int ptmi_debug_occluded(ptmi_ctx *ctx, uint32_t n, const float *o3, const float *d3,
                        const float *dist, uint8_t *occluded);
/* Host-only (no context, no device): builds the traversal image ptmi_upload_scene would build and reports on it.
 * out[0] wide nodes of the rebuilt hierarchy (0: the tree is walked as uploaded), [1] leaves, [2] its depth,
 * [3] quantised nodes (0: none), [4] dwords of the leaf stream, [5] quantised child boxes that do NOT contain the exact
 * box they stand for (must be 0), [6] mean relative growth of box surface area by the quantisation, [7] structural
 * mismatches (must be 0): leaf headers or triangle records of the stream that differ from the uploaded leaf box / triangles,
 * inner boxes of the rebuilt hierarchy that are not the exact union of their children's boxes, nodes not reached once. */
int ptmi_debug_image_stats(const ptmi_triangle *triangles, uint32_t n_triangles,
                           const ptmi_bvh_node *bvh_nodes, uint32_t n_nodes, double out[8]);
/* Host-only (no context, no device): builds the traversal image ptmi_upload_scene would build under opt's `leaves`, `leaf_tris` and
 * `keep_reference_tree` (NULL: the defaults) and copies it out, for tests and tools that walk it on the CPU. Every buffer may be NULL
 * (call once for the sizes). wnodes16: n_wnodes x 16 floats (csrc/pt_device.h: two child boxes, two references); qnodes8: n_wnodes x 8
 * words (only when info->quantised; numbered top-of-tree first, not like wnodes16); tripos12: n_tris x 12 floats — (v0, w), e1, e2 with
 * w = bits(original triangle index) and the triangles in LEAF order when leaves_used = 2; leafbox8: n_triangles x 8 floats, the box of
 * the reference leaf that lists each (original) triangle (leaves_used = 2 only). */
typedef struct ptmi_image_info {
    uint32_t leaves_used, n_wnodes, n_tris, root_ref, depth, n_leaves, max_leaf_tris, quantised;
    float root_min[3], root_max[3];   /* the box tested first (leaves_used = 2: padded) */
    float pad, safe_origin;           /* leaves_used = 2: what every box was padded by; the origin distance the padding is good for */
    float q_origin[3], q_scale[3];
    uint32_t ref_depth;               /* levels of the uploaded tree */
} ptmi_image_info;
int ptmi_debug_build_image(const ptmi_triangle *triangles, uint32_t n_triangles, const ptmi_bvh_node *bvh_nodes, uint32_t n_nodes,
                           const ptmi_options *opt, ptmi_image_info *info, float *wnodes16, uint32_t *qnodes8, float *tripos12,
                           float *leafbox8);
/* arithmetic-contract probe: out[i] = op(a[i], b[i], c[i]) evaluated on the device.
 * ops: 0 a/b, 1 sqrt(a), 2 fma(a,b,c), 3 min(a,b), 4 max(a,b), 5 sin(a), 6 cos(a),
 *      7 pow5(a), 8 f32(u32 bits of a), 9 u32(a) as bits, 10 a - trunc(a), 11 tan(a), 12 1/a (the kernels' short form) */
int ptmi_debug_math(ptmi_ctx *ctx, int op, uint32_t n, const float *a, const float *b,
                    const float *c, float *out);
/* The kernels compute 1/x and sqrt(x) with short instruction sequences where the operand's magnitude is within
 * [2^-100, 2^100] and with the compiler's IEEE expansions elsewhere (csrc/pt_math.h). This runs both over ALL 2^32 float bit
 * patterns on the device and reports how many inputs give different bits (two NaNs count as equal) and the smallest such
 * pattern: which = 0: 1/x; 1: sqrt(x); 2: the 1/x of the triangle test, whose result is only used for |x| >= 1e-6.
 * The arithmetic contract (correctly rounded results, as the CPU oracle's '/' and sqrtf) holds iff all three report 0. */
int ptmi_debug_exact_math(ptmi_ctx *ctx, int which, uint64_t *n_different, uint32_t *first_different);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
