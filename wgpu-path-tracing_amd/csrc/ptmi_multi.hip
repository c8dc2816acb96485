// ptmi_multi.hip — several GPUs of one node behind the C ABI (include/ptmi.h, "several GPUs").
//
// The caller is the reference's frame loop (src/renderer/renderer.ts:415-454): one host thread driving one Renderer. Here the
// Renderer owns N device contexts; the frame's rows are dealt out as interleaved strips (DevBand, pt_device.h), every device
// traces and accumulates its own rows — pixels and RNG streams are independent (pt.wgsl:719, :753-761), so there is no
// collective on the data path — and ptmi_multi_gather assembles the frame on device 0:
//
//     k_pack_rows (each device: its rows -> one contiguous buffer)   on that device's stream
//     ncclGather  (one group call, root = device 0; RCCL over xGMI)  on the same streams
//     k_unpack_rows (device 0: buffer r -> the rows of device r)     on device 0's stream
//
// Equal counts per rank are what ncclGather takes, so every device sends rows_max x width float4 (the last round of strips may
// leave some devices a strip short; the padding is never unpacked). RCCL is loaded with dlopen when the first handle is
// created: a process that renders on one device never maps it.
#include "ptmi.h"
#include "pt_device.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_multi_create_err;

// the RCCL entry points this file uses, resolved once (rccl.h supplies the types only)
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Gather = reinterpret_cast<decltype(Gather)>(sym("ncclGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather || !GetErrorString) {
            dlclose(lib); lib = nullptr; return false;
        }
        return true;
    }
} g_rccl;

constexpr uint32_t kStripRows = 4;      // measured on Cornell: every N-th 4-row strip is an even sample of the picture

}  // namespace

struct ptmi_multi {
    std::vector<ptmi_ctx *> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;                      // empty: loopback copies
    bool loopback = false;
    ptmi_options opt{};                                // as given by the caller (tile_strip 0 = automatic)
    uint32_t W = 0, H = 0, strip = kStripRows;
    size_t rows_max = 0;                               // rows of the largest share
    size_t share_bytes = 0;                            // what every d_send[i] holds (d_recv: N of them): rows_max x W float4 when allocated
    std::vector<float4 *> d_send;                      // per device: its packed rows
    float4 *d_recv = nullptr;                          // device 0: N shares
    std::vector<hipEvent_t> ev;                        // loopback: device r's share is packed
    std::vector<hipEvent_t> ev_done;                   // device r's rendering is done (recorded on its stream when a gather starts)
    std::vector<hipEvent_t> ev_copied;                 // loopback: device r's share has been copied out of d_send[r] (recorded on device 0's stream)
    std::vector<char> copied_recorded;
    hipEvent_t g0 = nullptr, g1 = nullptr;             // around the last gather on device 0's stream
    bool gather_timed = false;
    uint64_t dispatched = 0, gathered = 0;             // dispatch calls so far / included in device 0's frame
    mutable std::string err;
};

namespace {

int mfail(const ptmi_multi *m, int code, const char *fmt, ...) {
    char buf[640];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (m) m->err = buf; else g_multi_create_err = buf;
    return code;
}
// a failed call on device i: carry its message
int cfail(const ptmi_multi *m, int i, int rc, const char *what) {
    return mfail(m, rc, "%s on device %d (ordinal %d): %s", what, i, m->dev[i], ptmi_last_error(m->ctx[i]));
}
#define MHIP(m, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return mfail((m), PTMI_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define MNCCL(m, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return mfail((m), PTMI_E_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); } while (0)

// strip height for n devices: kStripRows when the frame is a whole number of rounds of that, else the largest smaller height that
// is (2160 rows over 8 devices: 3); ragged frames work with any height, a whole number of rounds only keeps the shares equal
uint32_t auto_strip(uint32_t H, uint32_t n) {
    if (n <= 1) return kStripRows;
    for (uint32_t s = kStripRows; s > 0; s--) if (H % (s * n) == 0) return s;
    return kStripRows;
}

ptmi_options options_of(const ptmi_multi *m, int i) {
    ptmi_options o = m->opt;
    const uint32_t n = (uint32_t)m->ctx.size();
    o.tile_y0 = 0; o.tile_y1 = 0;
    o.tile_parts = n > 1 ? n : 0; o.tile_part = n > 1 ? (uint32_t)i : 0; o.tile_strip = m->strip;
    return o;
}

void free_buffers(ptmi_multi *m) {
    for (size_t i = 0; i < m->d_send.size(); i++)
        if (m->d_send[i]) { (void)hipSetDevice(m->dev[i]); (void)hipFree(m->d_send[i]); m->d_send[i] = nullptr; }
    if (m->d_recv) { (void)hipSetDevice(m->dev[0]); (void)hipFree(m->d_recv); m->d_recv = nullptr; }
    m->rows_max = 0; m->share_bytes = 0;
}

// options of every context + the gather's buffers for the current size and strip height
int configure(ptmi_multi *m) {
    const int n = (int)m->ctx.size();
    m->strip = m->opt.tile_strip ? m->opt.tile_strip : auto_strip(m->H ? m->H : 1, (uint32_t)n);
    for (int i = 0; i < n; i++) {
        const ptmi_options o = options_of(m, i);
        int rc = ptmi_set_options(m->ctx[i], &o);
        if (rc) return cfail(m, i, rc, "ptmi_set_options");
    }
    if (m->W == 0 || n == 1) return PTMI_OK;
    size_t rows_max = 0;
    for (int i = 0; i < n; i++) rows_max = std::max<size_t>(rows_max, pt_band_of(options_of(m, i), m->W, m->H).rows);
    // the buffers are sized in BYTES (rows_max x W float4): a resize that keeps the height but widens the frame needs new ones
    const size_t share = rows_max * m->W * sizeof(float4);
    if (rows_max == m->rows_max && share == m->share_bytes && m->d_recv) return PTMI_OK;
    int rc = ptmi_multi_synchronize(m);
    if (rc) return rc;
    free_buffers(m);
    for (int i = 0; i < n; i++) {
        MHIP(m, hipSetDevice(m->dev[i]));
        MHIP(m, hipMalloc(&m->d_send[i], std::max<size_t>(share, 16)));
        MHIP(m, hipMemset(m->d_send[i], 0, std::max<size_t>(share, 16)));
    }
    MHIP(m, hipSetDevice(m->dev[0]));
    MHIP(m, hipMalloc(&m->d_recv, std::max<size_t>(share * n, 16)));
    m->rows_max = rows_max; m->share_bytes = share;
    std::fill(m->copied_recorded.begin(), m->copied_recorded.end(), 0);
    return PTMI_OK;
}

}  // namespace

extern "C" {

const char *ptmi_multi_last_error(const ptmi_multi *m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

int ptmi_multi_create(int n, const int *ordinals, uint32_t flags, ptmi_multi **out) {
    if (!out) return mfail(nullptr, PTMI_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 1 || n > 64) return mfail(nullptr, PTMI_E_INVALID, "n_devices %d not in 1..64", n);
    if (flags & ~(uint32_t)PTMI_MULTI_LOOPBACK) return mfail(nullptr, PTMI_E_INVALID, "unknown flags 0x%x", flags);
    ptmi_multi *m = new ptmi_multi();
    m->loopback = (flags & PTMI_MULTI_LOOPBACK) != 0;
    for (int i = 0; i < n; i++) m->dev.push_back(ordinals ? ordinals[i] : i);
    if (!m->loopback)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++)
                if (m->dev[i] == m->dev[j]) {
                    const int d = m->dev[i]; delete m;
                    return mfail(nullptr, PTMI_E_INVALID, "device ordinal %d is listed twice (only PTMI_MULTI_LOOPBACK lets one device stand in for several)", d);
                }
    for (int i = 0; i < n; i++) {
        ptmi_ctx *c = nullptr;
        int rc = ptmi_create(m->dev[i], &c);
        if (rc) {
            mfail(nullptr, rc, "ptmi_create(%d): %s", m->dev[i], ptmi_last_error(nullptr));
            ptmi_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    m->d_send.assign(n, nullptr);
    ptmi_get_options(m->ctx[0], &m->opt);
    m->opt.tile_strip = 0;
    if (!m->loopback && n >= 1) {
        // one communicator per device, one process (ncclCommInitAll). A single device goes through RCCL too: its gather is the
        // degenerate collective, and the un-sharded bits must come out of it unchanged.
        if (!g_rccl.load()) { mfail(nullptr, PTMI_E_UNSUPPORTED, "%s", g_rccl.err.c_str()); ptmi_multi_destroy(m); return PTMI_E_UNSUPPORTED; }
        m->comm.assign(n, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n, m->dev.data());
        if (r != ncclSuccess) {
            m->comm.clear();
            mfail(nullptr, PTMI_E_HIP, "ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
            ptmi_multi_destroy(m);
            return PTMI_E_HIP;
        }
    }
    bool ok = true;
    m->ev.assign(n, nullptr); m->ev_copied.assign(n, nullptr); m->ev_done.assign(n, nullptr); m->copied_recorded.assign(n, 0);
    for (int i = 0; i < n && ok; i++) {
        ok = hipSetDevice(m->dev[i]) == hipSuccess && hipEventCreateWithFlags(&m->ev[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipSetDevice(m->dev[0]) == hipSuccess && hipEventCreateWithFlags(&m->ev_copied[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipSetDevice(m->dev[0]) == hipSuccess && hipEventCreate(&m->g0) == hipSuccess && hipEventCreate(&m->g1) == hipSuccess;
    if (!ok) { mfail(nullptr, PTMI_E_HIP, "event creation failed"); ptmi_multi_destroy(m); return PTMI_E_HIP; }
    int rc = configure(m);
    if (rc) { g_multi_create_err = m->err; ptmi_multi_destroy(m); return rc; }
    *out = m;
    return PTMI_OK;
}

int ptmi_multi_destroy(ptmi_multi *m) {
    if (!m) return PTMI_E_INVALID;
    for (ptmi_ctx *c : m->ctx) (void)ptmi_synchronize(c);
    for (ncclComm_t c : m->comm) if (c) (void)g_rccl.CommDestroy(c);
    free_buffers(m);
    for (size_t i = 0; i < m->ev.size(); i++) if (m->ev[i]) { (void)hipSetDevice(m->dev[i]); (void)hipEventDestroy(m->ev[i]); }
    for (size_t i = 0; i < m->ev_done.size(); i++) if (m->ev_done[i]) { (void)hipSetDevice(m->dev[i]); (void)hipEventDestroy(m->ev_done[i]); }
    for (size_t i = 0; i < m->ev_copied.size(); i++) if (m->ev_copied[i]) { (void)hipSetDevice(m->dev[0]); (void)hipEventDestroy(m->ev_copied[i]); }
    if (m->g0) (void)hipEventDestroy(m->g0);
    if (m->g1) (void)hipEventDestroy(m->g1);
    for (ptmi_ctx *c : m->ctx) (void)ptmi_destroy(c);
    delete m;
    return PTMI_OK;
}

int ptmi_multi_count(const ptmi_multi *m) { return m ? (int)m->ctx.size() : 0; }
ptmi_ctx *ptmi_multi_context(ptmi_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int ptmi_multi_upload_scene(ptmi_multi *m, const ptmi_triangle *tris, uint32_t nt, const ptmi_material *mats, uint32_t nm,
                            const ptmi_bvh_node *nodes, uint32_t nn, const ptmi_light *lights, uint32_t nl) {
    if (!m) return PTMI_E_INVALID;
    // validation, the hierarchy and its images are built ONCE on the host (under device 0's options: all devices share them) and copied
    // to every device — the 1 M-triangle scene costs one build, not N (round 3: N x 196 ms)
    int rc = PTMI_OK;
    PtPrepared *p = pt_prepare_scene(m->ctx[0], tris, nt, mats, nm, nodes, nn, lights, nl, &rc);
    if (!p) return cfail(m, 0, rc, "ptmi_upload_scene (prepare)");
    for (size_t i = 0; i < m->ctx.size() && rc == PTMI_OK; i++) {
        rc = pt_install_scene(m->ctx[i], p);
        if (rc) rc = cfail(m, (int)i, rc, "ptmi_upload_scene (install)");
    }
    pt_free_prepared(p);
    return rc;
}

int ptmi_multi_upload_atlas(ptmi_multi *m, const void *texels, uint32_t w, uint32_t h, int fmt) {
    if (!m) return PTMI_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        int rc = ptmi_upload_atlas(m->ctx[i], texels, w, h, fmt);
        if (rc) return cfail(m, (int)i, rc, "ptmi_upload_atlas");
    }
    return PTMI_OK;
}

int ptmi_multi_resize(ptmi_multi *m, uint32_t w, uint32_t h) {
    if (!m) return PTMI_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        int rc = ptmi_resize(m->ctx[i], w, h);
        if (rc) return cfail(m, (int)i, rc, "ptmi_resize");
    }
    m->W = w; m->H = h;
    m->dispatched = m->gathered = 0;
    return configure(m);
}

int ptmi_multi_set_options(ptmi_multi *m, const ptmi_options *o) {
    if (!m || !o) return PTMI_E_INVALID;
    if (o->tile_y0 != 0 || o->tile_y1 != 0) return mfail(m, PTMI_E_INVALID, "tile_y0 / tile_y1 must be 0: the rows are dealt out by the library");
    // The strip height decides which device owns which rows: changing it while frames are being accumulated would hand rows that
    // hold k frames to a device whose copy of them holds none. It takes effect with the next ptmi_multi_resize / _write_output.
    const uint32_t new_strip = o->tile_strip ? o->tile_strip : auto_strip(m->H ? m->H : 1, (uint32_t)m->ctx.size());
    if (m->dispatched != 0 && new_strip != m->strip)
        return mfail(m, PTMI_E_STATE, "tile_strip %u -> %u while %llu dispatches are accumulated: call ptmi_multi_resize or ptmi_multi_write_output first",
                     m->strip, new_strip, (unsigned long long)m->dispatched);
    const ptmi_options old = m->opt;
    m->opt = *o;
    int rc = configure(m);
    if (rc) { m->opt = old; (void)configure(m); }
    return rc;
}
int ptmi_multi_get_options(const ptmi_multi *m, ptmi_options *o) {
    if (!m || !o) return PTMI_E_INVALID;
    *o = m->opt; o->tile_strip = m->strip; o->tile_parts = (uint32_t)m->ctx.size(); o->tile_part = 0;
    return PTMI_OK;
}

int ptmi_multi_dispatch(ptmi_multi *m, const ptmi_camera *cam, uint32_t n_frames) {
    if (!m) return PTMI_E_INVALID;
    const size_t n = m->ctx.size();
    // One ptmi_dispatch of 64 frames is ~45 launches and ~110 event calls: 1.15 ms of host time (profiles/r04_multi.json). Enqueued
    // in turn from one thread, device i would start i x that after device 0 — 8 ms at N = 8 against 16 ms of device time per step of
    // configs[4] — so every device gets its own enqueuing thread for the call (contexts are independent: own device, own streams;
    // a context is still only ever touched by one thread at a time).
    std::vector<int> rcs(n, PTMI_OK);
    if (n > 1) {
        std::vector<std::thread> pool;
        pool.reserve(n - 1);
        for (size_t i = 1; i < n; i++) pool.emplace_back([&, i] { rcs[i] = ptmi_dispatch(m->ctx[i], cam, n_frames); });
        rcs[0] = ptmi_dispatch(m->ctx[0], cam, n_frames);
        for (std::thread &t : pool) t.join();
    } else {
        rcs[0] = ptmi_dispatch(m->ctx[0], cam, n_frames);
    }
    for (size_t i = 0; i < n; i++) if (rcs[i]) return cfail(m, (int)i, rcs[i], "ptmi_dispatch");
    m->dispatched++;
    return PTMI_OK;
}

int ptmi_multi_gather(ptmi_multi *m) {
    if (!m) return PTMI_E_INVALID;
    const int n = (int)m->ctx.size();
    if (n == 1 && m->comm.empty()) { m->gathered = m->dispatched; return PTMI_OK; }
    if (m->W == 0) return mfail(m, PTMI_E_STATE, "no output buffer (ptmi_multi_resize)");
    if (n == 1) {
        // one device through RCCL: the degenerate gather of its whole frame onto itself, then copied back — the frame must come
        // out of the collective unchanged (the N = 1 test of the RCCL leg on a one-GPU box)
        const size_t frame_bytes = (size_t)m->W * m->H * sizeof(float4);
        if (!m->d_recv || m->rows_max != m->H || m->share_bytes != frame_bytes) {
            int rc = ptmi_multi_synchronize(m); if (rc) return rc;
            free_buffers(m);
            MHIP(m, hipSetDevice(m->dev[0]));
            MHIP(m, hipMalloc(&m->d_send[0], frame_bytes));
            MHIP(m, hipMalloc(&m->d_recv, frame_bytes));
            m->rows_max = m->H; m->share_bytes = frame_bytes;
        }
    }
    const size_t share_f4 = m->rows_max * m->W;
    hipStream_t s0 = pt_ctx_stream(m->ctx[0]);
    // the timed region (ptmi_multi_gather_ms) starts when EVERY device has rendered its rows: pack, gather, unpack — not the wait for
    // the slowest device, which is the dispatch's time
    for (int i = 1; i < n; i++) {
        MHIP(m, hipSetDevice(m->dev[i]));
        MHIP(m, hipEventRecord(m->ev_done[i], pt_ctx_stream(m->ctx[i])));
        MHIP(m, hipSetDevice(m->dev[0]));
        MHIP(m, hipStreamWaitEvent(s0, m->ev_done[i], 0));
    }
    MHIP(m, hipSetDevice(m->dev[0]));
    MHIP(m, hipEventRecord(m->g0, s0));
    std::vector<DevBand> bands(n);
    for (int i = 0; i < n; i++) {
        bands[i] = pt_band_of(options_of(m, i), m->W, m->H);
        MHIP(m, hipSetDevice(m->dev[i]));
        // loopback: the previous gather's copy out of d_send[i] (on device 0's stream) must be done before it is packed again
        if (m->comm.empty() && i > 0 && m->copied_recorded[i]) MHIP(m, hipStreamWaitEvent(pt_ctx_stream(m->ctx[i]), m->ev_copied[i], 0));
        if (bands[i].rows) pt_launch_pack_rows(pt_ctx_stream(m->ctx[i]), pt_ctx_cus(m->ctx[i]) * 8, bands[i], pt_ctx_output(m->ctx[i]), m->d_send[i]);
    }
    if (!m->comm.empty()) {
        MNCCL(m, g_rccl.GroupStart());
        for (int i = 0; i < n; i++) {
            ncclResult_t r = g_rccl.Gather(m->d_send[i], i == 0 ? m->d_recv : nullptr, share_f4 * 4, ncclFloat, 0, m->comm[i], pt_ctx_stream(m->ctx[i]));
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return mfail(m, PTMI_E_HIP, "ncclGather (device %d) failed: %s", i, g_rccl.GetErrorString(r)); }
        }
        MNCCL(m, g_rccl.GroupEnd());
    } else {
        // loopback: device r's share is copied into slot r of device 0's receive buffer once it is packed
        for (int i = 1; i < n; i++) {
            MHIP(m, hipSetDevice(m->dev[i]));
            MHIP(m, hipEventRecord(m->ev[i], pt_ctx_stream(m->ctx[i])));
            MHIP(m, hipSetDevice(m->dev[0]));
            MHIP(m, hipStreamWaitEvent(s0, m->ev[i], 0));
            MHIP(m, hipMemcpyPeerAsync(m->d_recv + (size_t)i * share_f4, m->dev[0], m->d_send[i], m->dev[i], share_f4 * sizeof(float4), s0));
            MHIP(m, hipEventRecord(m->ev_copied[i], s0));
            m->copied_recorded[i] = 1;
        }
    }
    MHIP(m, hipSetDevice(m->dev[0]));
    // device 0's own rows are already in place (RCCL delivers a copy of them into slot 0; the single-device case unpacks that
    // copy, so that the frame really went through the collective)
    for (int i = m->comm.empty() || n > 1 ? 1 : 0; i < n; i++)
        if (bands[i].rows) pt_launch_unpack_rows(s0, pt_ctx_cus(m->ctx[0]) * 8, bands[i], m->d_recv + (size_t)i * share_f4, pt_ctx_output(m->ctx[0]));
    MHIP(m, hipEventRecord(m->g1, s0));
    MHIP(m, hipGetLastError());
    m->gather_timed = true;
    m->gathered = m->dispatched;
    return PTMI_OK;
}

int ptmi_multi_synchronize(ptmi_multi *m) {
    if (!m) return PTMI_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        int rc = ptmi_synchronize(m->ctx[i]);
        if (rc) return cfail(m, (int)i, rc, "ptmi_synchronize");
    }
    return PTMI_OK;
}

int ptmi_multi_throttle(ptmi_multi *m, uint32_t max_in_flight, uint32_t *in_flight) {
    if (!m) return PTMI_E_INVALID;
    uint32_t worst = 0;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        uint32_t n = 0;
        int rc = ptmi_throttle(m->ctx[i], max_in_flight, &n);
        if (rc) return cfail(m, (int)i, rc, "ptmi_throttle");
        worst = std::max(worst, n);
    }
    if (in_flight) *in_flight = worst;
    return PTMI_OK;
}

int ptmi_multi_read_output(ptmi_multi *m, float *dst, size_t n_floats) {
    if (!m || !dst) return PTMI_E_INVALID;
    if (m->gathered != m->dispatched) { int rc = ptmi_multi_gather(m); if (rc) return rc; }
    int rc = ptmi_multi_synchronize(m);
    if (rc) return rc;
    rc = ptmi_read_output(m->ctx[0], dst, n_floats);
    return rc ? cfail(m, 0, rc, "ptmi_read_output") : PTMI_OK;
}

int ptmi_multi_write_output(ptmi_multi *m, const float *src, size_t n_floats) {
    if (!m || !src) return PTMI_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        int rc = ptmi_write_output(m->ctx[i], src, n_floats);
        if (rc) return cfail(m, (int)i, rc, "ptmi_write_output");
    }
    m->dispatched = m->gathered = 0;          // every device holds the whole frame again: the rows may be dealt out anew
    return PTMI_OK;
}

int ptmi_multi_blit(ptmi_multi *m, float *dst_f32, size_t n_floats, uint8_t *dst_rgba8, size_t n_bytes) {
    if (!m) return PTMI_E_INVALID;
    if (m->gathered != m->dispatched) { int rc = ptmi_multi_gather(m); if (rc) return rc; }
    int rc = ptmi_multi_synchronize(m);
    if (rc) return rc;
    rc = ptmi_blit(m->ctx[0], dst_f32, n_floats, dst_rgba8, n_bytes);
    return rc ? cfail(m, 0, rc, "ptmi_blit") : PTMI_OK;
}

int ptmi_multi_get_stats(ptmi_multi *m, ptmi_stats *out) {
    if (!m || !out) return PTMI_E_INVALID;
    ptmi_stats sum;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        ptmi_stats s;
        int rc = ptmi_get_stats(m->ctx[i], &s);
        if (rc) return cfail(m, (int)i, rc, "ptmi_get_stats");
        if (i == 0) { sum = s; continue; }
        sum.paths += s.paths; sum.segments += s.segments; sum.shadow_rays += s.shadow_rays; sum.shadow_traced += s.shadow_traced;
        for (int b = 0; b < 64; b++) sum.segments_by_bounce[b] += s.segments_by_bounce[b];
        sum.gpu_ms = std::max(sum.gpu_ms, s.gpu_ms); sum.extend_ms = std::max(sum.extend_ms, s.extend_ms);
        sum.shade_ms = std::max(sum.shade_ms, s.shade_ms); sum.shadow_ms = std::max(sum.shadow_ms, s.shadow_ms);
        sum.raygen_ms = std::max(sum.raygen_ms, s.raygen_ms); sum.compact_ms = std::max(sum.compact_ms, s.compact_ms);
        sum.accumulate_ms = std::max(sum.accumulate_ms, s.accumulate_ms);
        sum.upload_ms = std::max(sum.upload_ms, s.upload_ms);
        sum.verify_failed += s.verify_failed;
    }
    *out = sum;
    return PTMI_OK;
}

int ptmi_multi_reset_stats(ptmi_multi *m) {
    if (!m) return PTMI_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        int rc = ptmi_reset_stats(m->ctx[i]);
        if (rc) return cfail(m, (int)i, rc, "ptmi_reset_stats");
    }
    return PTMI_OK;
}

int ptmi_multi_gather_ms(ptmi_multi *m, double *ms) {
    if (!m || !ms) return PTMI_E_INVALID;
    *ms = -1.0;
    if (!m->gather_timed) return PTMI_OK;
    int rc = ptmi_multi_synchronize(m);
    if (rc) return rc;
    float f = 0.0f;
    MHIP(m, hipSetDevice(m->dev[0]));
    MHIP(m, hipEventElapsedTime(&f, m->g0, m->g1));
    *ms = f;
    return PTMI_OK;
}

}  // extern "C"
