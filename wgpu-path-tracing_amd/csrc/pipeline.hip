// pipeline.hip — the stream kernels around traversal and shading:
//   raygen      camera rays + RNG seeding          (reference: src/shader/pt.wgsl:714-750)
//   compact     ordered compaction of survivors    (no reference counterpart: the megakernel
//               keeps dead lanes idle; here ballot words -> popcount prefix -> scatter)
//   accumulate  clamp + running mean in frame order (pt.wgsl:751-761)
#include "pt_device.h"
#include "pt_math.h"

namespace {

constexpr int BLOCK = 256;

// pt.wgsl:719-750
PT_DEV void camera_ray(const ptmi_camera &cam, uint32_t x, uint32_t y, uint32_t frame, v3 &org, v3 &dir,
                       uint32_t &rng) {
    rng = rng_seed(x, y, frame);
    float jx = rng_f(rng), jy = rng_f(rng);
    float px = (float)x + jx, py = (float)y + jy;
    float uvx = (px / (float)cam.width) * 2.0f - 1.0f;
    float uvy = (py / (float)cam.height) * 2.0f - 1.0f;
    float th = tan1(cam.fov * 0.5f);
    v3 fw = mk3(cam.forward[0], cam.forward[1], cam.forward[2]);
    v3 rt = mk3(cam.right[0], cam.right[1], cam.right[2]);
    v3 up = mk3(cam.up[0], cam.up[1], cam.up[2]);
    v3 pos = mk3(cam.position[0], cam.position[1], cam.position[2]);
    v3 a = scale3(scale3(scale3(rt, uvx), th), cam.aspect);
    v3 b = scale3(scale3(up, uvy), th);
    dir = normalize3(add3(add3(fw, a), b));
    org = pos;
    if (cam.aperture > 0.0f) {
        v3 focal = madd3(dir, cam.focus_distance, pos);
        float r = sqrt1(rng_f(rng)) * cam.aperture;
        float theta = rng_f(rng) * 2.0f * PT_PI;
        float st, ct; sincos1(theta, st, ct);
        v3 off = madd3(up, r * st, scale3(rt, r * ct));
        org = add3(pos, off);
        dir = normalize3(sub3(focal, org));
    }
}

// The throughput of pt.wgsl:639 is not stored: `shade` knows it is (1, 1, 1) at bounce 0 and writes D.w / C from then on.
PT_DEV void init_path(DevPaths P, uint32_t p, v3 o, v3 d, uint32_t rng) {
    P.O[p] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
    P.D[p] = make_float4(d.x, d.y, d.z, 0.0f);
    P.stL(p, 0.0f, 0.0f, 0.0f);                        // pt.wgsl:640
}

// path id = frame_in_batch * band_pixels + local_row * width + x (local rows: DevBand::row_of). The bounce-0 queue is the identity and is
// not materialised: extend / shade / compact take a null queue as "slot i holds path i".
__global__ __launch_bounds__(BLOCK) void k_raygen(ptmi_camera cam, DevBand band, uint32_t frame0, uint32_t n_frames,
                                                  DevPaths P, uint32_t *__restrict__ count_out) {
    const uint32_t npix = band.rows * band.width;
    const uint32_t total = npix * n_frames;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = total;
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < total; p += gridDim.x * BLOCK) {
        uint32_t k = p / npix, pix = p - k * npix;
        uint32_t y = band.row_of(pix / band.width), x = pix % band.width;
        v3 o, d; uint32_t rng;
        camera_ray(cam, x, y, frame0 + k, o, d, rng);
        init_path(P, p, o, d, rng);
    }
}

__global__ __launch_bounds__(BLOCK) void k_raygen_list(ptmi_camera cam, uint32_t n, const uint32_t *xs,
                                                       const uint32_t *ys, const uint32_t *frames, DevPaths P) {
    uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    v3 o, d; uint32_t rng;
    camera_ray(cam, xs[p], ys[p], frames[p], o, d, rng);
    init_path(P, p, o, d, rng);
}

// ---- ordered compaction ------------------------------------------------------
// `shade` leaves one ballot word per 64 queue slots. A tile = 1024 words = 65536 slots,
// one 1024-thread workgroup.
//   k_tile_sums : per-tile popcount totals (+ the statistics counters)
//   k_scatter   : every tile sums the totals of the tiles before it (a few hundred at most),
//                 scans its own 1024 popcounts (wave shuffles + 16 wave totals in LDS), then each
//                 wave walks its 64 words: word j's mask and base offset are read from lane j,
//                 lane L keeps slot 64*w+L iff bit L is set, at base + popcount(bits below L).
// The next queue is therefore the surviving path ids in unchanged (ascending) order, and its
// length lands in next_count — no host round trip.
// Measured on config 1 (Msamples/s, same box): 256 words per tile 8 898, 512: 9 133, 1024: 9 160 — with 133 M slots a launch
// of 8 100 four-wave workgroups costs more than the prefix over fewer, larger tiles.
#ifndef PT_TILE_WORDS
#define PT_TILE_WORDS 1024
#endif
constexpr int TILE_WORDS = PT_TILE_WORDS;   // ballot words (x 64 queue slots) per tile = threads per workgroup
constexpr int TILE_WAVES = TILE_WORDS / 64;

__global__ __launch_bounds__(TILE_WORDS) void k_tile_sums(const uint32_t *__restrict__ count_ptr,
                                                          const uint64_t *__restrict__ alive,
                                                          const uint64_t *__restrict__ shadow,
                                                          uint32_t *__restrict__ tile_sums,
                                                          uint32_t *__restrict__ shadow_tile_sums,
                                                          unsigned long long *__restrict__ stats, uint32_t bounce) {
    __shared__ uint32_t wsum[TILE_WAVES], wssum[TILE_WAVES];
    const uint32_t count = *count_ptr;
    const uint32_t nwords = (count + 63u) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // atomics: two batches may be in flight on two lanes
        atomicAdd(&stats[0], (unsigned long long)count); atomicAdd(&stats[8 + bounce], (unsigned long long)count);
    }
    if (blockIdx.x * TILE_WORDS >= nwords) return;
    const uint32_t w = blockIdx.x * TILE_WORDS + threadIdx.x;
    uint32_t c = 0, sc = 0;
    if (w < nwords) { c = (uint32_t)__popcll(alive[w]); if (shadow) sc = (uint32_t)__popcll(shadow[w]); }
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off); sc += __shfl_down(sc, off); }
    if ((threadIdx.x & 63u) == 0u) { wsum[threadIdx.x >> 6] = c; wssum[threadIdx.x >> 6] = sc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, ts = 0;
        for (int i = 0; i < TILE_WAVES; i++) { t += wsum[i]; ts += wssum[i]; }
        tile_sums[blockIdx.x] = t;
        shadow_tile_sums[blockIdx.x] = ts;
        if (ts) { atomicAdd(&stats[1], (unsigned long long)ts); atomicAdd(&stats[2], (unsigned long long)ts); }   // shadow rays, and those traced (integers: order-free)
    }
}

PT_DEV void scatter_tile(uint32_t tile, const uint32_t *__restrict__ count_ptr, const uint32_t *__restrict__ queue,
                         const uint64_t *__restrict__ alive, const uint32_t *__restrict__ tile_sums,
                         uint32_t *__restrict__ next_queue, uint32_t *__restrict__ next_count) {
    __shared__ uint32_t wtot[TILE_WAVES];
    __shared__ uint32_t tile_base;
    const uint32_t count = *count_ptr;
    const uint32_t nwords = (count + 63u) >> 6;
    const uint32_t ntiles = (nwords + TILE_WORDS - 1) / TILE_WORDS;
    if (tile >= ntiles) {
        if (ntiles == 0 && tile == 0 && threadIdx.x == 0) *next_count = 0;
        return;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // totals of the tiles in front of this one
    uint32_t pre = 0;
    for (uint32_t t = threadIdx.x; t < tile; t += TILE_WORDS) pre += tile_sums[t];
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off);
    if (lane == 0) wtot[wave] = pre;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int i = 0; i < TILE_WAVES; i++) t += wtot[i]; tile_base = t; }
    __syncthreads();
    const uint32_t base0 = tile_base;
    __syncthreads();
    // exclusive scan of this tile's popcounts
    const uint32_t w = tile * TILE_WORDS + threadIdx.x;
    const uint64_t m = w < nwords ? alive[w] : 0ull;
    const uint32_t c = (uint32_t)__popcll(m);
    uint32_t inc = c;
    for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += v; }
    if (lane == 63u) wtot[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t i = 0; i < wave; i++) wbase += wtot[i];
    const uint32_t my_off = base0 + wbase + inc - c;                  // offset of this lane's word
    if (tile == ntiles - 1 && threadIdx.x == TILE_WORDS - 1) *next_count = my_off + c;
    // cooperative scatter: the wave walks its 64 words
    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
    const uint32_t w0 = tile * TILE_WORDS + wave * 64u;
    for (uint32_t j = 0; j < 64u; j++) {
        const uint32_t jlo = __shfl(mlo, (int)j), jhi = __shfl(mhi, (int)j), jbase = __shfl(my_off, (int)j);
        const uint64_t jm = ((uint64_t)jhi << 32) | jlo;
        if (jm == 0ull) continue;
        if ((jm >> lane) & 1ull) {
            const uint32_t below = (uint32_t)__popcll(jm & ((1ull << lane) - 1ull));
            const uint32_t slot = (w0 + j) * 64u + lane;
            next_queue[jbase + below] = queue ? queue[slot] : slot;
        }
    }
}

__global__ __launch_bounds__(TILE_WORDS) void k_scatter(const uint32_t *__restrict__ count_ptr,
                                                        const uint32_t *__restrict__ queue,
                                                        const uint64_t *__restrict__ alive,
                                                        const uint32_t *__restrict__ tile_sums,
                                                        uint32_t *__restrict__ next_queue,
                                                        uint32_t *__restrict__ next_count) {
    scatter_tile(blockIdx.x, count_ptr, queue, alive, tile_sums, next_queue, next_count);
}
// both compactions of a bounce in one launch: workgroups [0, tiles) compact the survivors into the next queue,
// [tiles, 2 tiles) the emitted records into the shadow index list
__global__ __launch_bounds__(TILE_WORDS) void k_scatter2(uint32_t tiles, const uint32_t *__restrict__ count_ptr,
                                                         const uint32_t *__restrict__ queue,
                                                         const uint64_t *__restrict__ alive, const uint64_t *__restrict__ shadow,
                                                         const uint32_t *__restrict__ tile_sums, const uint32_t *__restrict__ shadow_sums,
                                                         uint32_t *__restrict__ next_queue, uint32_t *__restrict__ next_count,
                                                         uint32_t *__restrict__ shadow_queue, uint32_t *__restrict__ shadow_count) {
    if (blockIdx.x < tiles) scatter_tile(blockIdx.x, count_ptr, queue, alive, tile_sums, next_queue, next_count);
    else scatter_tile(blockIdx.x - tiles, count_ptr, nullptr, shadow, shadow_sums, shadow_queue, shadow_count);
}

// pt.wgsl:751-761 for the batch's frames in ascending order
__global__ __launch_bounds__(BLOCK) void k_accumulate(DevBand band, uint32_t frame0, uint32_t n_frames,
                                                      const float *__restrict__ L, uint32_t l_stride, float4 *__restrict__ out) {
    const uint32_t npix = band.rows * band.width;
    for (uint32_t pix = blockIdx.x * BLOCK + threadIdx.x; pix < npix; pix += gridDim.x * BLOCK) {
        const size_t oi = (size_t)band.row_of(pix / band.width) * band.width + pix % band.width;
        float4 acc = out[oi];
        for (uint32_t k = 0; k < n_frames; k++) {
            const size_t li = (size_t)k * npix + pix;
            rgb_sc l;
            if (l_stride == 4u) { const float4 v = reinterpret_cast<const float4 *>(L)[li]; l = rgb_sc{v.x, v.y, v.z}; }
            else l = reinterpret_cast<const rgb_sc *>(L)[li];
            float cx = min1(l.x, 2.5f), cy = min1(l.y, 2.5f), cz = min1(l.z, 2.5f);
            uint32_t frame = frame0 + k;
            if (frame > 0u) {
                float t = 1.0f / (float)(frame + 1u);
                cx = mix1(acc.x, cx, t); cy = mix1(acc.y, cy, t); cz = mix1(acc.z, cz, t);
            }
            acc = make_float4(cx, cy, cz, 0.0f);
        }
        out[oi] = acc;
    }
}

// ---- multi-GPU gather (ptmi_multi_gather): a device's rows (DevBand: the strips part, part + parts, ...) <-> one contiguous
// buffer of band.rows x width float4, local row l of the buffer = frame row band.row_of(l)
__global__ __launch_bounds__(BLOCK) void k_pack_rows(DevBand band, const float4 *__restrict__ frame, float4 *__restrict__ packed) {
    const uint32_t n = band.rows * band.width;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        packed[i] = frame[(size_t)band.row_of(i / band.width) * band.width + i % band.width];
}
__global__ __launch_bounds__(BLOCK) void k_unpack_rows(DevBand band, const float4 *__restrict__ packed, float4 *__restrict__ frame) {
    const uint32_t n = band.rows * band.width;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        frame[(size_t)band.row_of(i / band.width) * band.width + i % band.width] = packed[i];
}

// ---- presentation: src/shader/blit.wgsl:43-155 (exposure 2^1, AgX, look, EOTF, gamma 1/2.2) -------------
// Not on the parity-exact path: log2 / pow are the device's own (accurate) library functions; the
// test bar is |gpu - oracle| <= 2e-5 per channel and equal 8-bit codes on >= 99.9 % of pixels.
PT_DEV v3 agx_contrast(v3 x) {                                   // blit.wgsl:54-65
    auto f = [](float v) {
        float x2 = v * v, x4 = x2 * x2;
        return 15.5f * x4 * x2 - 40.14f * x4 * v + 31.96f * x4 - 6.868f * x2 * v + 0.4298f * x2 + 0.1191f * v - 0.00232f;
    };
    return mk3(f(x.x), f(x.y), f(x.z));
}
PT_DEV v3 tone_map(v3 c) {                                       // blit.wgsl:133-145
    c = scale3(c, 2.0f);                                         // exposureAdjust: color * exp2(1.0)
    // agx(): inset matrix (column-major mat3x3f * vec3), log2 encoding, sigmoid approximation (:67-86)
    v3 r = mk3(0.842479062253094f * c.x + 0.0784335999999992f * c.y + 0.0792237451477643f * c.z,
               0.0423282422610123f * c.x + 0.878468636469772f * c.y + 0.0791661274605434f * c.z,
               0.0423756549057051f * c.x + 0.0784336f * c.y + 0.879142973793104f * c.z);
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    r = mk3(min1(max1(__builtin_log2f(r.x), min_ev), max_ev), min1(max1(__builtin_log2f(r.y), min_ev), max_ev),
            min1(max1(__builtin_log2f(r.z), min_ev), max_ev));
    r = mk3((r.x - min_ev) / (max_ev - min_ev), (r.y - min_ev) / (max_ev - min_ev), (r.z - min_ev) / (max_ev - min_ev));
    r = agx_contrast(r);
    // agxLook(): default look, slope = power = sat = 1 (:102-114)
    float luma = r.x * 0.2126f + r.y * 0.7152f + r.z * 0.0722f;
    r = mk3(luma + (__builtin_powf(r.x, 1.0f) - luma), luma + (__builtin_powf(r.y, 1.0f) - luma),
            luma + (__builtin_powf(r.z, 1.0f) - luma));
    // agxEotf(): outset matrix, then ^2.2 (:88-100)
    v3 e = mk3(1.19687900512017f * r.x - 0.0980208811401368f * r.y - 0.0990297440797205f * r.z,
               -0.0528968517574562f * r.x + 1.15190312990417f * r.y - 0.0989611768448433f * r.z,
               -0.0529716355144438f * r.x - 0.0980434501171241f * r.y + 1.15107367264116f * r.z);
    return mk3(__builtin_powf(e.x, 2.2f), __builtin_powf(e.y, 2.2f), __builtin_powf(e.z, 2.2f));
}
// one thread per canvas pixel (i, j), j from the top; fragmentMain, blit.wgsl:147-155
__global__ __launch_bounds__(BLOCK) void k_blit(uint32_t W, uint32_t H, const float4 *__restrict__ color,
                                                float4 *__restrict__ out_f32, uint32_t *__restrict__ out_rgba8) {
    const uint32_t n = W * H;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) {
        const uint32_t i = k % W, j = k / W;
        const float uvx = ((float)i + 0.5f) / (float)W, uvy = ((float)j + 0.5f) / (float)H;
        const uint32_t x = f2u(uvx * (float)(W - 1u));
        const uint32_t y = f2u((1.0f - uvy) * (float)(H - 1u));
        const float4 c4 = color[(size_t)y * W + x];
        v3 c = tone_map(mk3(c4.x, c4.y, c4.z));
        const float g = 1.0f / 2.2f;
        c = mk3(__builtin_powf(c.x, g), __builtin_powf(c.y, g), __builtin_powf(c.z, g));   // gammaCorrect
        if (out_f32) out_f32[k] = make_float4(c.x, c.y, c.z, 1.0f);
        if (out_rgba8) {
            auto q = [](float v) { v = min1(max1(v, 0.0f), 1.0f); return (uint32_t)(v * 255.0f + 0.5f); };   // NaN -> 0
            out_rgba8[k] = q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | 0xFF000000u;
        }
    }
}

__global__ void k_math(int op, uint32_t n, const float *a, const float *b, const float *c, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, z = c ? c[i] : 0.0f, r = 0.0f, t;
    switch (op) {
    case 0: r = x / y; break;
    case 1: r = sqrt1(x); break;
    case 2: r = fma1(x, y, z); break;
    case 3: r = min1(x, y); break;
    case 4: r = max1(x, y); break;
    case 5: sincos1(x, r, t); break;
    case 6: sincos1(x, t, r); break;
    case 7: r = pow5(x); break;
    case 8: r = (float)__float_as_uint(x); break;
    case 9: r = __uint_as_float(f2u(x)); break;
    case 10: r = x - __builtin_truncf(x); break;
    case 11: r = tan1(x); break;
    case 12: r = rcp1(x); break;
    default: break;
    }
    out[i] = r;
}

// The short forms of 1/x and sqrt(x) (pt_math.h) against the compiler's IEEE expansions, on every float there is:
// which = 0: rcp1(x) vs 1.0f / x; 1: sqrt1(x) vs sqrtf(x); 2: rcp1_above_eps(x) vs 1.0f / x for |x| >= PT_EPS or NaN.
// out[0] = inputs whose results differ in any bit (two NaNs count as equal), out[1] = the smallest such bit pattern.
__global__ void k_exact_math(int which, unsigned long long *out) {
    unsigned long long bad = 0, first = ~0ull;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        float got, ref;
        if (which == 1) { got = sqrt1(x); ref = __builtin_sqrtf(x); }
        else if (which == 2) { if (__builtin_fabsf(x) < PT_EPS) continue; got = rcp1_above_eps(x); ref = 1.0f / x; }
        else { got = rcp1(x); ref = 1.0f / x; }
        if (__float_as_uint(got) != __float_as_uint(ref) && !(got != got && ref != ref)) { bad++; first = i < first ? i : first; }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

}  // namespace

void pt_launch_exact_math(hipStream_t s, int which, unsigned long long *out) {
    hipLaunchKernelGGL(k_exact_math, dim3(256 * 16), dim3(256), 0, s, which, out);
}
void pt_launch_raygen(hipStream_t s, int blocks, const ptmi_camera &cam, DevBand band, uint32_t frame0,
                      uint32_t n_frames, DevPaths p, uint32_t *count_out) {
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(BLOCK), 0, s, cam, band, frame0, n_frames, p, count_out);
}
void pt_launch_raygen_list(hipStream_t s, const ptmi_camera &cam, uint32_t n, const uint32_t *xs,
                           const uint32_t *ys, const uint32_t *frames, DevPaths p) {
    hipLaunchKernelGGL(k_raygen_list, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, cam, n, xs, ys, frames, p);
}
void pt_launch_compact(hipStream_t s, int tiles, const uint32_t *queue, const uint32_t *count,
                       const uint64_t *alive_mask, const uint64_t *shadow_mask, uint32_t *tile_sums,
                       uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_queue, uint32_t *shadow_count,
                       unsigned long long *stats, uint32_t bounce, int do_scatter) {
    uint32_t *shadow_sums = tile_sums + tiles;
    hipLaunchKernelGGL(k_tile_sums, dim3(tiles), dim3(TILE_WORDS), 0, s, count, alive_mask, shadow_mask, tile_sums,
                       shadow_sums, stats, bounce);
#ifndef PT_NO_SCATTER2
    if (do_scatter && shadow_mask) {
        hipLaunchKernelGGL(k_scatter2, dim3(2 * tiles), dim3(TILE_WORDS), 0, s, (uint32_t)tiles, count, queue, alive_mask, shadow_mask,
                           tile_sums, shadow_sums, next_queue, next_count, shadow_queue, shadow_count);
        return;
    }
#endif
    if (do_scatter)
        hipLaunchKernelGGL(k_scatter, dim3(tiles), dim3(TILE_WORDS), 0, s, count, queue, alive_mask, tile_sums,
                           next_queue, next_count);
    if (shadow_mask)    // slots (not path ids) of the emitted shadow records, ascending
        hipLaunchKernelGGL(k_scatter, dim3(tiles), dim3(TILE_WORDS), 0, s, count, (const uint32_t *)nullptr,
                           shadow_mask, shadow_sums, shadow_queue, shadow_count);
}
void pt_launch_accumulate(hipStream_t s, int blocks, DevBand band, uint32_t frame0, uint32_t n_frames,
                          const float *L, uint32_t l_stride, float4 *out) {
    hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(BLOCK), 0, s, band, frame0, n_frames, L, l_stride, out);
}
void pt_launch_pack_rows(hipStream_t s, int blocks, DevBand band, const float4 *frame, float4 *packed) {
    hipLaunchKernelGGL(k_pack_rows, dim3(blocks), dim3(BLOCK), 0, s, band, frame, packed);
}
void pt_launch_unpack_rows(hipStream_t s, int blocks, DevBand band, const float4 *packed, float4 *frame) {
    hipLaunchKernelGGL(k_unpack_rows, dim3(blocks), dim3(BLOCK), 0, s, band, packed, frame);
}
void pt_launch_blit(hipStream_t s, int blocks, uint32_t W, uint32_t H, const float4 *color, float4 *out_f32,
                    uint32_t *out_rgba8) {
    hipLaunchKernelGGL(k_blit, dim3(blocks), dim3(BLOCK), 0, s, W, H, color, out_f32, out_rgba8);
}
void pt_launch_math(hipStream_t s, int op, uint32_t n, const float *a, const float *b, const float *c, float *out) {
    hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, s, op, n, a, b, c, out);
}

uint32_t pt_compact_tile_slots(void) { return TILE_WORDS * 64u; }
