/*
 * ptmi_napi.c — thin N-API (Node >= 12, N-API 4) addon over the C ABI of include/ptmi.h.
 *
 * One JS function per C function; scene / camera blobs travel as ArrayBuffers (or typed-array
 * views) in the exact WGSL layouts the reference writes with device.queue.writeBuffer
 * (src/renderer/renderer.ts:242-355, :403-413). A non-zero ptmi status becomes a JS Error
 * carrying ptmi_last_error(). No rendering logic lives here.
 *
 * build: oracle-free, see Makefile next to this file
 *   gcc -shared -fPIC -I/usr/include/node -I../../../include ptmi_napi.c -L../../lib -lptmi
 */
#define NAPI_VERSION 4
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ptmi.h"
#include "ptmi_scene.h"

#define NAPI_OK(env, call)                                                        \
    do {                                                                          \
        if ((call) != napi_ok) {                                                  \
            napi_throw_error((env), NULL, "N-API call failed: " #call);           \
            return NULL;                                                          \
        }                                                                         \
    } while (0)

static napi_value throw_ptmi(napi_env env, ptmi_ctx *ctx, int rc, const char *what) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s failed (%d): %s", what, rc, ptmi_last_error(ctx));
    napi_throw_error(env, "PTMI", buf);
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static ptmi_ctx *get_ctx(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a ptmi context handle");
        return NULL;
    }
    return (ptmi_ctx *)p;
}

/* ArrayBuffer or TypedArray/DataView -> (pointer, byte length); null/undefined -> (NULL, 0) */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *len) {
    napi_valuetype t;
    bool is;
    *data = NULL; *len = 0;
    if (napi_typeof(env, v, &t) == napi_ok && (t == napi_null || t == napi_undefined)) return 1;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type tt; size_t n; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &tt, &n, data, &ab, &off) != napi_ok) return 0;
        static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *len = n * esz[tt];
        return 1;
    }
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab; size_t off;
        return napi_get_dataview_info(env, v, len, data, &ab, &off) == napi_ok;
    }
    napi_throw_type_error(env, NULL, "expected an ArrayBuffer or a typed array");
    return 0;
}

static uint32_t get_u32_prop(napi_env env, napi_value obj, const char *name, uint32_t dflt) {
    napi_value v; bool has = false; uint32_t out = dflt;
    if (napi_has_named_property(env, obj, name, &has) == napi_ok && has &&
        napi_get_named_property(env, obj, name, &v) == napi_ok)
        napi_get_value_uint32(env, v, &out);
    return out;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    int32_t dev = 0;
    NAPI_OK(env, napi_get_value_int32(env, argv[0], &dev));
    ptmi_ctx *ctx = NULL;
    int rc = ptmi_create(dev, &ctx);
    if (rc) return throw_ptmi(env, NULL, rc, "ptmi_create");
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, ctx, NULL, NULL, &ext));
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptmi_destroy(ctx);
    return NULL;
}

static napi_value js_upload_scene(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p[4]; size_t n[4];
    static const size_t stride[4] = {sizeof(ptmi_triangle), sizeof(ptmi_material), sizeof(ptmi_bvh_node), sizeof(ptmi_light)};
    for (int i = 0; i < 4; i++) {
        if (!get_bytes(env, argv[1 + i], &p[i], &n[i])) return NULL;
        if (n[i] % stride[i]) { napi_throw_range_error(env, NULL, "blob length is not a multiple of its element size"); return NULL; }
    }
    int rc = ptmi_upload_scene(ctx, (const ptmi_triangle *)p[0], (uint32_t)(n[0] / stride[0]),
                               (const ptmi_material *)p[1], (uint32_t)(n[1] / stride[1]),
                               (const ptmi_bvh_node *)p[2], (uint32_t)(n[2] / stride[2]),
                               (const ptmi_light *)p[3], (uint32_t)(n[3] / stride[3]));
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_upload_scene");
    return NULL;
}

static napi_value js_upload_atlas(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n; uint32_t w = 0, h = 0; int32_t fmt = 0;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    napi_get_value_uint32(env, argv[2], &w); napi_get_value_uint32(env, argv[3], &h); napi_get_value_int32(env, argv[4], &fmt);
    if (p && n < (size_t)w * h * (fmt == PTMI_ATLAS_RGBA16F ? 8 : 16)) { napi_throw_range_error(env, NULL, "atlas buffer too small"); return NULL; }
    int rc = ptmi_upload_atlas(ctx, p, w, h, fmt);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_upload_atlas");
    return NULL;
}

static napi_value js_resize(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t w = 0, h = 0;
    napi_get_value_uint32(env, argv[1], &w); napi_get_value_uint32(env, argv[2], &h);
    int rc = ptmi_resize(ctx, w, h);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_resize");
    return NULL;
}

/* JS option names -> ptmi_options (unset properties keep their current values) */
static void read_options(napi_env env, napi_value obj, ptmi_options *o) {
    o->max_bounces = get_u32_prop(env, obj, "maxBounces", o->max_bounces);
    o->do_mis = get_u32_prop(env, obj, "doMis", o->do_mis);
    o->tile_y0 = get_u32_prop(env, obj, "tileY0", o->tile_y0);
    o->tile_y1 = get_u32_prop(env, obj, "tileY1", o->tile_y1);
    o->frames_per_batch = get_u32_prop(env, obj, "framesPerBatch", o->frames_per_batch);
    o->traversal = get_u32_prop(env, obj, "traversal", o->traversal);
    o->cull = get_u32_prop(env, obj, "cull", o->cull);
    o->timing = get_u32_prop(env, obj, "timing", o->timing);
    o->keep_reference_tree = get_u32_prop(env, obj, "keepReferenceTree", o->keep_reference_tree);
    o->tile_parts = get_u32_prop(env, obj, "tileParts", o->tile_parts);
    o->tile_part = get_u32_prop(env, obj, "tilePart", o->tile_part);
    o->tile_strip = get_u32_prop(env, obj, "tileStrip", o->tile_strip);
    o->perf_mode = get_u32_prop(env, obj, "perfMode", o->perf_mode);
    o->overlap = get_u32_prop(env, obj, "overlap", o->overlap);
    o->tree_builder = get_u32_prop(env, obj, "treeBuilder", o->tree_builder);
    o->leaves = get_u32_prop(env, obj, "leaves", o->leaves);
    o->leaf_tris = get_u32_prop(env, obj, "leafTris", o->leaf_tris);
}

static napi_value js_set_options(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptmi_options o;
    ptmi_get_options(ctx, &o);
    read_options(env, argv[1], &o);
    int rc = ptmi_set_options(ctx, &o);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_set_options");
    return NULL;
}

static napi_value js_dispatch(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n; uint32_t frames = 1;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    if (!p || n != sizeof(ptmi_camera)) { napi_throw_range_error(env, NULL, "camera blob must be 96 bytes"); return NULL; }
    napi_get_value_uint32(env, argv[2], &frames);
    ptmi_camera cam;
    memcpy(&cam, p, sizeof cam);
    int rc = ptmi_dispatch(ctx, &cam, frames);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_dispatch");
    return NULL;
}

static napi_value js_synchronize(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = ptmi_synchronize(ctx);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_synchronize");
    return NULL;
}

/* throttle(ctx, maxInFlight) -> dispatches still in flight (blocks until at most maxInFlight are) */
static napi_value js_throttle(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint32_t max = 0, n = 0;
    napi_get_value_uint32(env, argv[1], &max);
    int rc = ptmi_throttle(ctx, max, &n);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_throttle");
    napi_value v;
    NAPI_OK(env, napi_create_uint32(env, n, &v));
    return v;
}

/* readOutput(ctx, Float32Array dst) */
static napi_value js_read_output(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    int rc = ptmi_read_output(ctx, (float *)p, n / 4);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_read_output");
    return argv[1];
}

static napi_value js_write_output(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    int rc = ptmi_write_output(ctx, (const float *)p, n / 4);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_write_output");
    return NULL;
}

/* blit(ctx, Uint8Array dstRgba8) — the reference's blit pass (blit.wgsl) into an 8-bit canvas, row 0 = top */
static napi_value js_blit(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    if (!p) { napi_throw_type_error(env, NULL, "expected a Uint8Array of width*height*4 bytes"); return NULL; }
    /* the library writes width*height*4 bytes: a short (or stale, after resize()) array must not reach it */
    uint32_t w = 0, h = 0;
    int rc = ptmi_get_size(ctx, &w, &h);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_get_size");
    if (n != (size_t)w * h * 4) {
        char msg[128];
        snprintf(msg, sizeof msg, "blit: expected a Uint8Array of %zu bytes (%ux%ux4), got %zu", (size_t)w * h * 4, w, h, n);
        napi_throw_range_error(env, NULL, msg);
        return NULL;
    }
    rc = ptmi_blit(ctx, NULL, 0, (uint8_t *)p, n);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_blit");
    return argv[1];
}

static void set_num(napi_env env, napi_value obj, const char *k, double v) {
    napi_value n;
    if (napi_create_double(env, v, &n) == napi_ok) napi_set_named_property(env, obj, k, n);
}

static napi_value stats_object(napi_env env, const ptmi_stats *s) {
    napi_value o;
    NAPI_OK(env, napi_create_object(env, &o));
    set_num(env, o, "paths", (double)s->paths); set_num(env, o, "segments", (double)s->segments);
    set_num(env, o, "shadowRays", (double)s->shadow_rays); set_num(env, o, "frames", (double)s->frames);
    set_num(env, o, "dispatches", (double)s->dispatches); set_num(env, o, "gpuMs", s->gpu_ms);
    set_num(env, o, "extendMs", s->extend_ms); set_num(env, o, "shadeMs", s->shade_ms); set_num(env, o, "shadowMs", s->shadow_ms);
    set_num(env, o, "shadowTraced", (double)s->shadow_traced); set_num(env, o, "uploadMs", s->upload_ms);
    set_num(env, o, "bvhDepth", s->bvh_depth); set_num(env, o, "traversalUsed", s->traversal_used);
    set_num(env, o, "framesPerBatchUsed", s->frames_per_batch_used);
    set_num(env, o, "leavesUsed", s->leaves_used); set_num(env, o, "leafTrisUsed", s->leaf_tris_used);
    set_num(env, o, "extendVariant", s->extend_variant); set_num(env, o, "shadowVariant", s->shadow_variant);
    set_num(env, o, "verifyFailed", (double)s->verify_failed);
    return o;
}

static napi_value js_get_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptmi_stats s;
    int rc = ptmi_get_stats(ctx, &s);
    if (rc) return throw_ptmi(env, ctx, rc, "ptmi_get_stats");
    return stats_object(env, &s);
}

static napi_value js_reset_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptmi_reset_stats(ctx);
    return NULL;
}


/* ---- several devices behind one handle (include/ptmi.h ptmi_multi_*): the same calls, one per JS function ---- */
static napi_value throw_multi(napi_env env, ptmi_multi *m, int rc, const char *what) {
    char buf[768];
    snprintf(buf, sizeof buf, "%s failed (%d): %s", what, rc, ptmi_multi_last_error(m));
    napi_throw_error(env, "PTMI", buf);
    return NULL;
}
static ptmi_multi *get_multi(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a ptmi multi-device handle");
        return NULL;
    }
    return (ptmi_multi *)p;
}

/* multiCreate([ordinal, ...], flags) */
static napi_value js_multi_create(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    uint32_t n = 0, flags = 0;
    bool is_array = false;
    if (napi_is_array(env, argv[0], &is_array) != napi_ok || !is_array || napi_get_array_length(env, argv[0], &n) != napi_ok || n < 1 || n > 64) {
        napi_throw_type_error(env, NULL, "expected an array of 1..64 device ordinals");
        return NULL;
    }
    int dev[64];
    for (uint32_t i = 0; i < n; i++) {
        napi_value e; int32_t d = 0;
        NAPI_OK(env, napi_get_element(env, argv[0], i, &e));
        NAPI_OK(env, napi_get_value_int32(env, e, &d));
        dev[i] = d;
    }
    napi_get_value_uint32(env, argv[1], &flags);
    ptmi_multi *m = NULL;
    int rc = ptmi_multi_create((int)n, dev, flags, &m);
    if (rc) return throw_multi(env, NULL, rc, "ptmi_multi_create");
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, m, NULL, NULL, &ext));
    return ext;
}

static napi_value js_multi_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (m) ptmi_multi_destroy(m);
    return NULL;
}

static napi_value js_multi_upload_scene(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p[4]; size_t n[4];
    static const size_t stride[4] = {sizeof(ptmi_triangle), sizeof(ptmi_material), sizeof(ptmi_bvh_node), sizeof(ptmi_light)};
    for (int i = 0; i < 4; i++) {
        if (!get_bytes(env, argv[1 + i], &p[i], &n[i])) return NULL;
        if (n[i] % stride[i]) { napi_throw_range_error(env, NULL, "blob length is not a multiple of its element size"); return NULL; }
    }
    int rc = ptmi_multi_upload_scene(m, (const ptmi_triangle *)p[0], (uint32_t)(n[0] / stride[0]),
                                     (const ptmi_material *)p[1], (uint32_t)(n[1] / stride[1]),
                                     (const ptmi_bvh_node *)p[2], (uint32_t)(n[2] / stride[2]),
                                     (const ptmi_light *)p[3], (uint32_t)(n[3] / stride[3]));
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_upload_scene");
    return NULL;
}

static napi_value js_multi_upload_atlas(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p; size_t n; uint32_t w = 0, h = 0; int32_t fmt = 0;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    napi_get_value_uint32(env, argv[2], &w); napi_get_value_uint32(env, argv[3], &h); napi_get_value_int32(env, argv[4], &fmt);
    if (p && n < (size_t)w * h * (fmt == PTMI_ATLAS_RGBA16F ? 8 : 16)) { napi_throw_range_error(env, NULL, "atlas buffer too small"); return NULL; }
    int rc = ptmi_multi_upload_atlas(m, p, w, h, fmt);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_upload_atlas");
    return NULL;
}

static napi_value js_multi_resize(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    uint32_t w = 0, h = 0;
    napi_get_value_uint32(env, argv[1], &w); napi_get_value_uint32(env, argv[2], &h);
    int rc = ptmi_multi_resize(m, w, h);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_resize");
    return NULL;
}

static napi_value js_multi_set_options(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    ptmi_options o;
    ptmi_multi_get_options(m, &o);
    o.tile_parts = 0; o.tile_part = 0; o.tile_strip = 0;            /* dealt out by the library unless tileStrip says otherwise */
    read_options(env, argv[1], &o);
    int rc = ptmi_multi_set_options(m, &o);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_set_options");
    return NULL;
}

static napi_value js_multi_dispatch(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p; size_t n; uint32_t frames = 1;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    if (!p || n != sizeof(ptmi_camera)) { napi_throw_range_error(env, NULL, "camera blob must be 96 bytes"); return NULL; }
    napi_get_value_uint32(env, argv[2], &frames);
    ptmi_camera cam;
    memcpy(&cam, p, sizeof cam);
    int rc = ptmi_multi_dispatch(m, &cam, frames);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_dispatch");
    return NULL;
}

static napi_value js_multi_gather(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    int rc = ptmi_multi_gather(m);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_gather");
    return NULL;
}

static napi_value js_multi_synchronize(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    int rc = ptmi_multi_synchronize(m);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_synchronize");
    return NULL;
}

static napi_value js_multi_throttle(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    uint32_t max = 0, n = 0;
    napi_get_value_uint32(env, argv[1], &max);
    int rc = ptmi_multi_throttle(m, max, &n);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_throttle");
    napi_value v;
    NAPI_OK(env, napi_create_uint32(env, n, &v));
    return v;
}

static napi_value js_multi_read_output(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    int rc = ptmi_multi_read_output(m, (float *)p, n / 4);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_read_output");
    return argv[1];
}

static napi_value js_multi_write_output(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    int rc = ptmi_multi_write_output(m, (const float *)p, n / 4);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_write_output");
    return NULL;
}

static napi_value js_multi_blit(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[1], &p, &n)) return NULL;
    if (!p) { napi_throw_type_error(env, NULL, "expected a Uint8Array of width*height*4 bytes"); return NULL; }
    uint32_t w = 0, h = 0;
    int rc = ptmi_get_size(ptmi_multi_context(m, 0), &w, &h);
    if (rc) return throw_multi(env, m, rc, "ptmi_get_size");
    if (n != (size_t)w * h * 4) { napi_throw_range_error(env, NULL, "blit: the Uint8Array is not width*height*4 bytes"); return NULL; }
    rc = ptmi_multi_blit(m, NULL, 0, (uint8_t *)p, n);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_blit");
    return argv[1];
}

static napi_value js_multi_get_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (!m) return NULL;
    ptmi_stats s;
    int rc = ptmi_multi_get_stats(m, &s);
    if (rc) return throw_multi(env, m, rc, "ptmi_multi_get_stats");
    napi_value o = stats_object(env, &s);
    if (!o) return NULL;
    double ms = -1.0;
    if (ptmi_multi_gather_ms(m, &ms) == 0) set_num(env, o, "gatherMs", ms);
    set_num(env, o, "devices", ptmi_multi_count(m));
    return o;
}

static napi_value js_multi_reset_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptmi_multi *m = get_multi(env, argv[0]);
    if (m) ptmi_multi_reset_stats(m);
    return NULL;
}

/* ---- host-side scene preparation (include/ptmi_scene.h, libptmi_scene.so; no GPU involved) ---- */

/* buildBvh(trianglesArrayBuffer) -> { nodes: ArrayBuffer (48-B nodes), depth }; sorts the triangles in place
 * exactly like src/renderer/bvh.ts does */
static napi_value js_build_bvh(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    void *p; size_t n;
    if (!get_bytes(env, argv[0], &p, &n)) return NULL;
    if (!p || n % sizeof(ptmi_triangle)) { napi_throw_range_error(env, NULL, "expected a triangle blob (128-byte elements)"); return NULL; }
    uint32_t nt = (uint32_t)(n / sizeof(ptmi_triangle));
    uint32_t cap = ptmi_scene_bvh_node_bound(nt), count = 0, depth = 0;
    void *nodes = NULL;
    napi_value ab;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)cap * sizeof(ptmi_bvh_node), &nodes, &ab));
    int rc = ptmi_scene_build_bvh((ptmi_triangle *)p, nt, 4, 12, (ptmi_bvh_node *)nodes, cap, &count, &depth);
    if (rc) { napi_throw_error(env, "PTMI_SCENE", ptmi_scene_last_error()); return NULL; }
    /* hand back exactly `count` nodes */
    void *out = NULL;
    napi_value ab2, obj, d;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)count * sizeof(ptmi_bvh_node), &out, &ab2));
    memcpy(out, nodes, (size_t)count * sizeof(ptmi_bvh_node));
    NAPI_OK(env, napi_create_object(env, &obj));
    NAPI_OK(env, napi_create_uint32(env, depth, &d));
    napi_set_named_property(env, obj, "nodes", ab2);
    napi_set_named_property(env, obj, "depth", d);
    return obj;
}

/* emissiveLights(triangles, materials, punctualLights) -> ArrayBuffer of 48-B lights: the punctual ones first,
 * then one per emissive triangle in post-sort order (src/renderer/gpu.ts:121-138) */
static napi_value js_emissive_lights(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    void *pt, *pm, *pl; size_t nt, nm, nl;
    if (!get_bytes(env, argv[0], &pt, &nt) || !get_bytes(env, argv[1], &pm, &nm) || !get_bytes(env, argv[2], &pl, &nl)) return NULL;
    if (nt % sizeof(ptmi_triangle) || nm % sizeof(ptmi_material) || nl % sizeof(ptmi_light)) {
        napi_throw_range_error(env, NULL, "blob length is not a multiple of its element size"); return NULL;
    }
    uint32_t ntri = (uint32_t)(nt / sizeof(ptmi_triangle)), n0 = (uint32_t)(nl / sizeof(ptmi_light)), count = n0;
    uint32_t cap = n0 + ntri;
    void *tmp = NULL;
    napi_value ab;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)(cap ? cap : 1) * sizeof(ptmi_light), &tmp, &ab));
    if (n0) memcpy(tmp, pl, nl);
    if (ntri) {
        int rc = ptmi_scene_emissive_lights((const ptmi_triangle *)pt, ntri, (const ptmi_material *)pm,
                                            (uint32_t)(nm / sizeof(ptmi_material)), (ptmi_light *)tmp, cap, &count);
        if (rc) { napi_throw_error(env, "PTMI_SCENE", ptmi_scene_last_error()); return NULL; }
    }
    void *out = NULL;
    napi_value ab2;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)count * sizeof(ptmi_light), &out, &ab2));
    if (count) memcpy(out, tmp, (size_t)count * sizeof(ptmi_light));
    return ab2;
}

static napi_value js_abi_version(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value v;
    NAPI_OK(env, napi_create_int32(env, ptmi_abi_version(), &v));
    return v;
}

static napi_value init(napi_env env, napi_value exports) {
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"abiVersion", js_abi_version}, {"create", js_create}, {"destroy", js_destroy},
        {"uploadScene", js_upload_scene}, {"uploadAtlas", js_upload_atlas}, {"resize", js_resize},
        {"setOptions", js_set_options}, {"dispatch", js_dispatch}, {"synchronize", js_synchronize}, {"throttle", js_throttle}, {"multiThrottle", js_multi_throttle},
        {"readOutput", js_read_output}, {"writeOutput", js_write_output}, {"blit", js_blit}, {"getStats", js_get_stats},
        {"resetStats", js_reset_stats}, {"buildBvh", js_build_bvh}, {"emissiveLights", js_emissive_lights},
        {"multiCreate", js_multi_create}, {"multiDestroy", js_multi_destroy}, {"multiUploadScene", js_multi_upload_scene},
        {"multiUploadAtlas", js_multi_upload_atlas}, {"multiResize", js_multi_resize}, {"multiSetOptions", js_multi_set_options},
        {"multiDispatch", js_multi_dispatch}, {"multiGather", js_multi_gather}, {"multiSynchronize", js_multi_synchronize},
        {"multiReadOutput", js_multi_read_output}, {"multiWriteOutput", js_multi_write_output}, {"multiBlit", js_multi_blit},
        {"multiGetStats", js_multi_get_stats}, {"multiResetStats", js_multi_reset_stats},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "cannot register addon functions");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
