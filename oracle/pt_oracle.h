/*
 * pt_oracle.h — CPU oracle for the path-tracing hot path (TEST INFRASTRUCTURE).
 *
 * A scalar f32 restatement of the reference's compute shader
 * (reference: src/shader/pt.wgsl + src/shader/random.wgsl). It is the checker
 * the HIP path is compared against and the timed CPU baseline of bench.py.
 * It is NOT part of the product: nothing under wgpu-path-tracing_amd/ links,
 * loads or calls it. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or
 * fixtures for this path (its only test, src/spec/arr.test.ts, pins the partial
 * quicksort used by the BVH builder), and its WGSL cannot be executed in this
 * environment (no WebGPU implementation). This restatement is pinned only by
 * its own KATs: the integer RNG vectors of SURVEY.md Appendix B (derived with
 * Python integers, independent of this code), analytic intersection cases,
 * closed-form BSDF identities and energy tests (tests/test_oracle_*.py).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include "../include/ptmi_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { PTO_ATLAS_NONE = 0, PTO_ATLAS_RGBA16F = 1, PTO_ATLAS_RGBA32F = 2 };

typedef struct pto_scene {
    const ptmi_triangle *tris;   uint32_t n_tris;
    const ptmi_material *mats;   uint32_t n_mats;
    const ptmi_bvh_node *nodes;  uint32_t n_nodes;
    const ptmi_light    *lights; uint32_t n_lights;
    const void *atlas; uint32_t atlas_w, atlas_h; int32_t atlas_fmt;
} pto_scene;

typedef struct pto_options {
    uint32_t max_bounces;   /* pt.wgsl:5 MAX_BOUNCES (8) */
    uint32_t do_mis;        /* pt.wgsl:636 DO_MIS (1) */
    uint32_t y0, y1;        /* rows [y0,y1) rendered; y1 = 0 means height */
    uint32_t threads;       /* 0 = omp default */
} pto_options;

typedef struct pto_stats {
    uint64_t paths;         /* (pixel, frame) samples started */
    uint64_t segments;      /* bounce-loop iterations reaching sceneIntersect (pt.wgsl:643-644) */
    uint64_t shadow_rays;   /* shadow traversals (pt.wgsl:392/421/463) */
    uint64_t nodes_visited; /* BVH nodes popped, closest-hit + shadow */
    uint64_t tris_tested;   /* ray/triangle tests */
    uint64_t closest_hits;  /* segments that hit something */
    uint32_t max_stack;     /* peak traversal-stack occupancy */
    uint32_t threads;       /* threads used */
    double   seconds;       /* wall time of the frame loop */
} pto_stats;

/* 1 = built with PT_STRICT (literal transcription: IEEE divisions everywhere,
 * no fused multiply-add, libm sin/cos/tan/pow); 0 = the arithmetic contract
 * of DESIGN.md that the HIP kernels implement bit-for-bit. */
int pto_is_strict(void);

/* random.wgsl:3-12. state_io is advanced n times; outputs (each may be NULL):
 * states[i] = rngState after draw i, words[i] = the hashed u32, vals[i] = rand(). */
void pto_rand(uint32_t *state_io, uint32_t n, uint32_t *states, uint32_t *words, float *vals);
/* random.wgsl:3-5 */
uint32_t pto_seed(uint32_t x, uint32_t y, uint32_t frame);
/* random.wgsl:14-16 with the N-1 clamp of DESIGN.md (rand()==1.0 case) */
uint32_t pto_rand_int(uint32_t *state_io, uint32_t lo, uint32_t hi);

/* contract sin/cos (x >= 0); in strict builds this is sinf/cosf */
void pto_sincos(float x, float *s, float *c);

/* pt.wgsl:714-750: camera ray for pixels (xs[i], ys[i]) at frames[i].
 * o3/d3: n*3 floats; rng_out[i] = rngState after the 2 or 4 raygen draws. */
int pto_raygen(const ptmi_camera *cam, uint32_t n, const uint32_t *xs, const uint32_t *ys,
               const uint32_t *frames, float *o3, float *d3, uint32_t *rng_out);

/* pt.wgsl:248-296 closest hit with the reference's rules (DFS, no t cull, first
 * strictly smaller t wins). t[i] = -1 on miss, tri[i] = 0xFFFFFFFF. */
int pto_intersect(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                  float *t, uint32_t *tri, float *u, float *v, pto_stats *st);

/* Shadow predicate of pt.wgsl:394 (dist[i] < 0: directional, occluded iff t>0)
 * and pt.wgsl:423/465 (occluded iff t>0 && t < dist - 2e-6). */
int pto_occluded(const pto_scene *s, uint32_t n, const float *o3, const float *d3,
                 const float *dist, uint8_t *occluded, pto_stats *st);

/* pt.wgsl:712-762 for frames cam->frame_index .. +n_frames-1 in order.
 * out_rgba: W*H*4 floats (xyz = running mean, w = 0), read when frame > 0. */
int pto_render(const pto_scene *s, const ptmi_camera *cam, uint32_t n_frames,
               const pto_options *opt, float *out_rgba, pto_stats *st);

/* Every ray a render traces, with the traversal's result (for tools that replay real rays through another traversal):
 * rec9[9 i ..] = o.xyz, d.xyz, dist (0: closest-hit ray; < 0: shadow ray to a directional light; > 0: shadow ray, the light's
 * distance), t (-1: miss), tri (bits). Rows [opt->y0, opt->y1) x n_frames; at most max_rays are stored, in no particular order;
 * returns how many were traced. No image is written. */
uint64_t pto_render_tap(const pto_scene *s, const ptmi_camera *cam, uint32_t n_frames, const pto_options *opt,
                        float *rec9, uint64_t max_rays);

/* One path, with a per-bounce log for debugging parity failures.
 * log: max_bounces+1 records of 16 floats:
 *   [0..2] ray origin, [3..5] ray dir, [6..8] throughput, [9..11] radiance,
 *   [12] rng state (bits), [13] hit t, [14] hit tri (bits), [15] alive flag
 * taken at the top of each bounce; the last written record has alive = 0.
 * radiance3 = the unclamped result of trace(). Returns the number of records. */
int pto_trace_path(const pto_scene *s, const ptmi_camera *cam, uint32_t x, uint32_t y,
                   uint32_t frame, const pto_options *opt, float *radiance3, float *log16);

/* blit.wgsl:43-155 (presentation pass; tolerance-compared). rgba/out: W*H*4 floats, out row 0 = canvas top. */
void pto_blit(const float *rgba, uint32_t W, uint32_t H, float *out_rgba);

/* out[i] = op(a[i], b[i], c[i]) with the contract's scalar helpers; op codes as
 * ptmi_debug_math in include/ptmi.h */
void pto_math(int op, uint32_t n, const float *a, const float *b, const float *c, float *out);

/* per-function probes used by the analytic KAT tests */
void pto_eval_bsdf(const float albedo[3], float roughness, float metallic, float transmission,
                   float ior, const float n[3], const float v[3], const float l[3], int front,
                   float out4[4]);
float pto_distribution_ggx(const float n[3], const float h[3], float roughness);
float pto_power_heuristic(float nf, float fpdf, float ng, float gpdf);
void pto_cosine_direction(uint32_t *state_io, float out3[3]);
void pto_sample_ggx_normal(uint32_t *state_io, const float n[3], float roughness, float out3[3]);

#ifdef __cplusplus
}
#endif
#endif
