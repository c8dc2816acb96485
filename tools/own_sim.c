/* own_sim.c — CPU replay of the per-ray arithmetic of csrc/traverse_own.hip (TOOL / TEST INFRASTRUCTURE, not product code).
 *
 * Walks the traversal image ptmi_debug_build_image returns (the library's own leaves, padded boxes, exact or quantised nodes)
 * with exactly the kernels' operations — the fused slab test, Moller-Trumbore with the contract's fused dot / cross products and a
 * correctly rounded reciprocal, the distance cull, the (t, lowest original index) rule, the verification of the winner against its
 * reference leaf's box and the retrace over the uploaded tree — one ray at a time, in float. What it is for:
 *   * the gate of VERDICT round 3 item 1 (box-pair steps and triangle tests per ray, before any GPU time is spent),
 *   * counting, on 10^8 and more rays taken from real renders (oracle/pt_oracle.c pto_render_tap), how many results differ from
 *     the reference traversal's, without a GPU (tools/own_leaf_gate.py, tests/test_own_leaves_host.py).
 * It shares no code with the oracle; the reference traversal it falls back to (slow rays) is restated here from
 * src/shader/pt.wgsl:234-291 of the reference.
 * Build: gcc -O2 -fopenmp -mfma -ffp-contract=off -fno-fast-math -shared -fPIC -Iinclude -o tools/build/libown_sim.so tools/own_sim.c -lm */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "ptmi_layout.h"

#define REF_LEAF 0x80000000u
#define REF_NONE 0xFFFFFFFFu
#define OFF_BITS 26u
#define OFF_MASK ((1u << OFF_BITS) - 1u)
#define EPS 1e-6f

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
/* v_min_f32 / v_max_f32: NaN -> the other operand */
static inline float min1(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
static inline float max1(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }
static inline float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}

typedef struct own_sim_scene {
    const float *wn; uint32_t n_wn;                  /* 16 floats per wide node */
    const uint32_t *qn; float qo[3], qs[3];          /* 8 words per node, or NULL */
    const float *tp; uint32_t n_tp;                  /* 12 floats per triangle, leaf order, tp[3] = bits(original index) */
    const float *leafbox;                            /* 8 floats per ORIGINAL triangle */
    uint32_t root_ref; float root_min[3], root_max[3];
    float safe_origin, tri_safe_dsum;
    const ptmi_bvh_node *nodes; uint32_t n_nodes;    /* the tree as uploaded */
    const ptmi_triangle *tris; uint32_t n_tris;
} own_sim_scene;

/* pt.wgsl:234-245 with (bound - o) * (1/d) — the contract's slab test */
static int slab(const float *lo, const float *hi, v3 o, v3 inv, float *tmin) {
    float t1x = (lo[0] - o.x) * inv.x, t2x = (hi[0] - o.x) * inv.x;
    float t1y = (lo[1] - o.y) * inv.y, t2y = (hi[1] - o.y) * inv.y;
    float t1z = (lo[2] - o.z) * inv.z, t2z = (hi[2] - o.z) * inv.z;
    *tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= *tmin && tmax >= 0.0f;
}
static int slab_t(float t1x, float t2x, float t1y, float t2y, float t1z, float t2z, float *tmin) {
    *tmin = max1(max1(min1(t1x, t2x), min1(t1y, t2y)), min1(t1z, t2z));
    float tmax = min1(min1(max1(t1x, t2x), max1(t1y, t2y)), max1(t1z, t2z));
    return tmax >= *tmin && tmax >= 0.0f;
}
/* pt.wgsl:128-158 as csrc/pt_math.h tri_test_t evaluates it */
static float tri_test(const float *v0, const float *e1p, const float *e2p, v3 o, v3 d) {
    v3 e1 = V(e1p[0], e1p[1], e1p[2]), e2 = V(e2p[0], e2p[1], e2p[2]);
    v3 h = cross3(d, e2);
    float a = dot3(e1, h);
    float f = 1.0f / a;
    v3 sv = V(o.x - v0[0], o.y - v0[1], o.z - v0[2]);
    float u = f * dot3(sv, h);
    v3 q = cross3(sv, e1);
    float v = f * dot3(d, q);
    float t = f * dot3(e2, q);
    int reject = (fabsf(a) < EPS) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
    return (!reject && t > EPS) ? t : -1.0f;
}
static inline float cull_limit(float t) { return fma_(t, 1.001f, 1e-4f); }

/* the reference's own traversal over the uploaded tree (what `slow` rays get): left-first DFS, no cull, first strictly nearer wins */
static void ref_trace(const own_sim_scene *s, v3 o, v3 d, float *t_out, uint32_t *tri_out) {
    uint32_t stack[128]; int sp = 0;
    float best = -1.0f; uint32_t btri = REF_NONE; int has = 0;
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    *t_out = -1.0f; *tri_out = REF_NONE;
    if (!s->n_nodes) return;
    stack[sp++] = 0;
    while (sp) {
        const ptmi_bvh_node *n = &s->nodes[stack[--sp]];
        float tm;
        if (!slab(n->aabb_min, n->aabb_max, o, inv, &tm)) continue;
        if (n->triangle_count) {
            for (uint32_t k = 0; k < n->triangle_count; k++) {
                const ptmi_triangle *T = &s->tris[n->triangle_offset + k];
                float e1[3] = {T->v1[0] - T->v0[0], T->v1[1] - T->v0[1], T->v1[2] - T->v0[2]};
                float e2[3] = {T->v2[0] - T->v0[0], T->v2[1] - T->v0[1], T->v2[2] - T->v0[2]};
                float t = tri_test(T->v0, e1, e2, o, d);
                if (t > 0.0f && (t < best || !has)) { best = t; btri = n->triangle_offset + k; has = 1; }
            }
        } else if (sp + 2 <= 128) { stack[sp++] = n->right; stack[sp++] = n->left; }
    }
    *t_out = best; *tri_out = btri;
}

typedef struct { v3 inv, n, s, oo; } pre_t;

static void node_test(const own_sim_scene *s, int quant, uint32_t i, const pre_t *p, float *tl, float *tr, int *hl, int *hr,
                      uint32_t *lref, uint32_t *rref) {
    if (quant) {
        const uint32_t *l = s->qn + 8u * (size_t)i, *r = l + 4;
#define PL(w, hi16) ((float)((hi16) ? ((w) >> 16) : ((w) & 0xFFFFu)))
        *hl = slab_t(fma_(PL(l[0], 0), p->s.x, p->oo.x), fma_(PL(l[1], 1), p->s.x, p->oo.x),
                     fma_(PL(l[0], 1), p->s.y, p->oo.y), fma_(PL(l[2], 0), p->s.y, p->oo.y),
                     fma_(PL(l[1], 0), p->s.z, p->oo.z), fma_(PL(l[2], 1), p->s.z, p->oo.z), tl);
        *hr = slab_t(fma_(PL(r[0], 0), p->s.x, p->oo.x), fma_(PL(r[1], 1), p->s.x, p->oo.x),
                     fma_(PL(r[0], 1), p->s.y, p->oo.y), fma_(PL(r[2], 0), p->s.y, p->oo.y),
                     fma_(PL(r[1], 0), p->s.z, p->oo.z), fma_(PL(r[2], 1), p->s.z, p->oo.z), tr);
#undef PL
        *lref = l[3]; *rref = r[3];
    } else {
        const float *w = s->wn + 16u * (size_t)i;
        /* q0 = (Lmin.xyz, Lmax.x) q1 = (Lmax.yz, Rmin.xy) q2 = (Rmin.z, Rmax.xyz) */
        *hl = slab_t(fma_(w[0], p->inv.x, p->n.x), fma_(w[3], p->inv.x, p->n.x), fma_(w[1], p->inv.y, p->n.y), fma_(w[4], p->inv.y, p->n.y),
                     fma_(w[2], p->inv.z, p->n.z), fma_(w[5], p->inv.z, p->n.z), tl);
        *hr = slab_t(fma_(w[6], p->inv.x, p->n.x), fma_(w[9], p->inv.x, p->n.x), fma_(w[7], p->inv.y, p->n.y), fma_(w[10], p->inv.y, p->n.y),
                     fma_(w[8], p->inv.z, p->n.z), fma_(w[11], p->inv.z, p->n.z), tr);
        memcpy(lref, &w[12], 4); memcpy(rref, &w[13], 4);
    }
}

/* one ray. dist: 0 closest hit; < 0 any hit, directional; > 0 any hit up to dist - 2e-6.
 * out: t (closest) or 1 / 0 (any hit: occluded), tri = the winner's original index; flags: 1 slow from the start, 2 winner failed its
 * reference leaf's box (retraced), 4 leaf list overflow (cannot happen); counts: box-pair steps, leaves opened, triangles tested */
static void trace_one(const own_sim_scene *s, int own, int quant, int cull, int deferred, v3 o, v3 d, float dist,
                      float *t_out, uint32_t *tri_out, uint8_t *flags, uint32_t *counts) {
    const int any = dist != 0.0f;
    const float tlim = !any ? 0.0f : (dist < 0.0f ? NAN : dist - EPS * 2.0f);
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float lo = 0x1p-60f, hi = 0x1p60f;     /* traverse_own.hip: direction components within [2^-60, 2^60] */
    int regular = (fabsf(d.x) >= lo) & (fabsf(d.x) <= hi) & (fabsf(d.y) >= lo) & (fabsf(d.y) <= hi) & (fabsf(d.z) >= lo) & (fabsf(d.z) <= hi);
    int bounded = (fabsf(d.x) + fabsf(d.y) + fabsf(d.z)) <= s->tri_safe_dsum;
    int near_o = (fabsf(o.x) <= s->safe_origin) & (fabsf(o.y) <= s->safe_origin) & (fabsf(o.z) <= s->safe_origin);
    if (!own) {                 /* the image over the reference's leaves (traverse.hip): only irregular rays walk the uploaded tree */
        regular = isfinite(inv.x) & isfinite(inv.y) & isfinite(inv.z) & (inv.x != 0.0f) & (inv.y != 0.0f) & (inv.z != 0.0f);
        near_o = 1;
    }
    int slow = !(regular & bounded & near_o);
    uint32_t steps = 0, leaves = 0, tris = 0;
    float best = INFINITY; uint32_t btri = REF_NONE; int occluded = 0;
    *flags = slow ? 1 : 0;
    if (!slow) {
        pre_t p;
        p.inv = inv; p.n = V(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
        p.s = V(s->qs[0] * inv.x, s->qs[1] * inv.y, s->qs[2] * inv.z);
        p.oo = V(fma_(s->qo[0], inv.x, p.n.x), fma_(s->qo[1], inv.y, p.n.y), fma_(s->qo[2], inv.z, p.n.z));
        float tm;
        float limit = (any && cull) ? cull_limit(tlim) : INFINITY;
        if (s->root_ref != REF_NONE && slab(s->root_min, s->root_max, o, inv, &tm)) {
            uint32_t stack[128]; int sp = 0;
            uint32_t filed[256]; int nf = 0;
            uint32_t cur = REF_NONE;
            if (s->root_ref & REF_LEAF) filed[nf++] = s->root_ref; else cur = s->root_ref;
            for (;;) {
                if (cur == REF_NONE && sp > 0) cur = stack[--sp];
                const int descent_over = cur == REF_NONE;
                /* leaves: at once (the best case of the kernels' scheduling) or after the whole descent (the worst) */
                if (nf && (!deferred || descent_over)) {
                    while (nf && !occluded) {
                        uint32_t ref = filed[--nf];
                        uint32_t first = ref & OFF_MASK, cnt = ((ref >> OFF_BITS) & 31u) + 1u;
                        leaves++;
                        for (uint32_t k = 0; k < cnt; k++) {
                            const float *T = s->tp + 12u * (size_t)(first + k);
                            tris++;
                            float t = tri_test(T, T + 4, T + 8, o, d);
                            uint32_t ti = first + k;
                            if (own) memcpy(&ti, &T[3], 4);
                            int hit = t > 0.0f;
                            if (any) {
                                int occ = hit && !(t >= tlim);
                                if (occ && !occluded) btri = ti;
                                occluded |= occ;
                            } else if (hit && (t < best || (t == best && ti < btri))) {
                                best = t; btri = ti;
                                if (cull) limit = cull_limit(t);
                            }
                        }
                    }
                    if (occluded) break;
                }
                if (descent_over) break;
                float tl, tr; int hl, hr; uint32_t lref, rref;
                steps++;
                if (own) node_test(s, quant, cur, &p, &tl, &tr, &hl, &hr, &lref, &rref);
                else {
                    const float *w = s->wn + 16u * (size_t)cur;
                    const float llo[3] = {w[0], w[1], w[2]}, lhi[3] = {w[3], w[4], w[5]}, rlo[3] = {w[6], w[7], w[8]}, rhi[3] = {w[9], w[10], w[11]};
                    hl = slab(llo, lhi, o, inv, &tl); hr = slab(rlo, rhi, o, inv, &tr);
                    memcpy(&lref, &w[12], 4); memcpy(&rref, &w[13], 4);
                }
                if (cull) { hl = hl && !(tl > limit); hr = hr && !(tr > limit); }
                int ll = (lref & REF_LEAF) != 0, rl = (rref & REF_LEAF) != 0;
                if (nf + 2 > 256) { *flags |= 4; break; }
                /* immediate mode tests the nearer leaf first */
                if ((hl && ll) && (hr && rl) && !deferred && tl <= tr) { filed[nf++] = rref; filed[nf++] = lref; }
                else { if (hl && ll) filed[nf++] = lref; if (hr && rl) filed[nf++] = rref; }
                int il = hl && !ll, ir = hr && !rl;
                if (il && ir) { int lf = tl <= tr; if (sp < 128) stack[sp++] = lf ? rref : lref; cur = lf ? lref : rref; }
                else if (il) cur = lref;
                else if (ir) cur = rref;
                else cur = REF_NONE;
            }
        }
        /* the winner must be a triangle the reference tests too */
        if (own && btri != REF_NONE) {
            const float *lb = s->leafbox + 8u * (size_t)btri;
            float tm2;
            if (!slab(lb, lb + 4, o, inv, &tm2)) { *flags |= 2; slow = 1; }
        }
    }
    if (slow) {
        float t; uint32_t tri;
        ref_trace(s, o, d, &t, &tri);
        if (any) { occluded = t > 0.0f && !(t >= tlim); btri = occluded ? tri : REF_NONE; }
        else { best = t > 0.0f ? t : INFINITY; btri = t > 0.0f ? tri : REF_NONE; }
    }
    if (any) { *t_out = occluded ? 1.0f : 0.0f; *tri_out = btri; }
    else { *t_out = btri == REF_NONE ? -1.0f : best; *tri_out = btri; }
    counts[0] = steps; counts[1] = leaves; counts[2] = tris;
}

/* own = 1: the image of ptmi_options.leaves = 2 (traverse_own.hip); own = 0: the hierarchy rebuilt over the reference's leaves
 * (leaves = 1, traverse.hip), walked with the contract's slab test — the baseline the gate compares with.
 * rec9: n rays as oracle/pt_oracle.c pto_render_tap leaves them (o, d, dist, the reference's t, tri).
 * out_t / out_tri / out_flags may be NULL. sums[0..2] += steps, leaves, triangles of the closest-hit rays, [3..5] of the shadow rays,
 * [6] closest-hit rays, [7] shadow rays, [8] closest-hit results that differ from rec9's (t bits or triangle), [9] shadow verdicts
 * that differ, [10] rays slow from the start, [11] rays retraced after a failed verification.
 * diff_idx: up to max_diff indices of differing rays (n_diff_out = how many were stored). */
void own_sim_run(const own_sim_scene *s, uint64_t n, const float *rec9, int own, int quant, int cull, int deferred,
                 float *out_t, uint32_t *out_tri, uint8_t *out_flags, uint64_t *sums, uint64_t *diff_idx, uint64_t max_diff,
                 uint64_t *n_diff_out) {
    uint64_t acc[12] = {0}, nd = 0;
#pragma omp parallel
    {
        uint64_t a[12] = {0};
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const float *r = rec9 + 9 * (size_t)i;
            float t; uint32_t tri, cnt[3]; uint8_t fl;
            trace_one(s, own, quant, cull, deferred, V(r[0], r[1], r[2]), V(r[3], r[4], r[5]), r[6], &t, &tri, &fl, cnt);
            const int any = r[6] != 0.0f;
            a[any ? 3 : 0] += cnt[0]; a[any ? 4 : 1] += cnt[1]; a[any ? 5 : 2] += cnt[2];
            a[any ? 7 : 6]++;
            a[10] += fl & 1; a[11] += (fl >> 1) & 1;
            uint32_t rtri; memcpy(&rtri, &r[8], 4);
            int differs;
            if (any) {
                const float tlim = r[6] < 0.0f ? NAN : r[6] - EPS * 2.0f;
                const int ref_occ = r[7] > 0.0f && !(r[7] >= tlim);
                differs = ref_occ != (t != 0.0f);
                a[9] += differs;
            } else {
                differs = memcmp(&t, &r[7], 4) != 0 || (r[7] > 0.0f && tri != rtri);
                a[8] += differs;
            }
            if (differs && diff_idx) {
                uint64_t k = __atomic_fetch_add(&nd, 1, __ATOMIC_RELAXED);
                if (k < max_diff) diff_idx[k] = (uint64_t)i;
            }
            if (out_t) out_t[i] = t;
            if (out_tri) out_tri[i] = tri;
            if (out_flags) out_flags[i] = fl;
        }
#pragma omp critical
        for (int k = 0; k < 12; k++) acc[k] += a[k];
    }
    if (sums) for (int k = 0; k < 12; k++) sums[k] += acc[k];
    if (n_diff_out) *n_diff_out = nd < max_diff ? nd : max_diff;
}
