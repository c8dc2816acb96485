"""GPU parity: the HIP kernels, called through the C ABI (libptmi.so), against the CPU
oracle on identical inputs. The bar is bit-exact — the oracle's contract build and the
kernels implement the same IEEE-754 operation order — for the integer RNG,
for (t, triangle, u, v) of every ray and for every output float."""
import numpy as np
import pytest

from ptmi import layout

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same_floats(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = (bits(a) != bits(b)) & ~both_nan
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} floats differ; first at {idx.tolist()}: "
                             f"gpu={a[tuple(idx[0])]!r} oracle={b[tuple(idx[0])]!r}")


# ---------------------------------------------------------------------------------
def test_math_contract(gpu_ctx, oracle):
    import ctypes
    rng = np.random.default_rng(7)
    n = 200_000
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 1e-45, 3.4e38, 2.5,
                        0.5, 4294967040.0, 4294967296.0, 16777217.0], np.float32)
    a = np.concatenate([special, rng.standard_normal(n).astype(np.float32) * 10,
                        (rng.random(n) * 6.2832).astype(np.float32)])
    b = np.concatenate([special[::-1], rng.standard_normal(n).astype(np.float32),
                        rng.standard_normal(n).astype(np.float32) * 1e-3])
    c = np.concatenate([special, rng.standard_normal(2 * n).astype(np.float32)])
    for op, name in enumerate(["div", "sqrt", "fma", "min", "max", "sin", "cos", "pow5", "u2f", "f2u", "frac", "tan"]):
        x = a
        if name in ("sin", "cos", "tan"):
            x = np.abs(a) % np.float32(6.3)
            x[np.isnan(x)] = 0.5
        if name == "sqrt":
            x = np.abs(a)
        ref = np.zeros_like(x)
        p = lambda q: q.ctypes.data_as(ctypes.c_void_p)
        oracle.L.pto_math(op, x.size, p(x), p(b), p(c), p(ref))
        got = gpu_ctx.debug_math(op, x, b, c)
        assert_same_floats(got, ref, f"math op {name}")


@pytest.mark.parametrize("aperture", [0.0, 0.001, 0.05])
def test_raygen_parity(gpu_ctx, oracle, aperture):
    cam = layout.make_camera(1920, 1080, aperture=aperture, focus_distance=2.8)
    rng = np.random.default_rng(3)
    n = 100_000
    xs = rng.integers(0, 1920, n).astype(np.uint32)
    ys = rng.integers(0, 1080, n).astype(np.uint32)
    fr = rng.integers(0, 512, n).astype(np.uint32)
    xs[:4], ys[:4], fr[:4] = [0, 1, 3, 1919], [0, 0, 2, 1079], [0, 0, 5, 63]      # SURVEY App. B seeds
    go, gd, grng = gpu_ctx.debug_raygen(cam, xs, ys, fr)
    oo, od, orng = oracle.raygen(cam, xs, ys, fr)
    assert np.array_equal(grng, orng)
    assert_same_floats(go, oo, "ray origin")
    assert_same_floats(gd, od, "ray direction")


def _test_rays(scene, n, seed):
    """Primary rays, interior random rays, axis-parallel and degenerate directions."""
    rng = np.random.default_rng(seed)
    o = np.empty((n, 3), np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lo = scene.nodes[0]["aabb_min"]
    hi = scene.nodes[0]["aabb_max"]
    o[:] = lo + (hi - lo) * rng.random((n, 3)).astype(np.float32)
    k = n // 4
    o[:k] = (0.0, 1.0, 2.8)                       # camera position: coherent primary-like rays
    d[:k, 2] = -np.abs(d[:k, 2]) - 1.0
    d[:k] /= np.linalg.norm(d[:k], axis=1, keepdims=True)
    d[k:k + 64] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 64)]          # exact zeros in d
    o[k + 64:k + 128] = scene.tris["v0"][rng.integers(0, len(scene.tris), 64)]  # origins on vertices
    return o, d.astype(np.float32)


@pytest.mark.parametrize("name", ["cornell", "feature_box", "cornell_spheres", "grid_1m"])
@pytest.mark.parametrize("cull", [1, 0])
@pytest.mark.parametrize("trav", ["global", "global_exact", "lds"])      # global = the quantised 32-byte image
def test_extend_parity(gpu_ctx, oracle, scene_factory, name, cull, trav):
    from ptmi import native
    sc = scene_factory(name)
    gpu_ctx.upload_scene(sc)
    mode = {"global": native.TRAVERSAL_GLOBAL, "global_exact": native.TRAVERSAL_GLOBAL_EXACT, "lds": native.TRAVERSAL_AUTO}[trav]
    gpu_ctx.set_options(cull=cull, traversal=mode)
    o, d = _test_rays(sc, 300_000, 11)
    gt, gtri, gu, gv = gpu_ctx.debug_intersect(o, d)
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    assert (ot > 0).mean() > 0.5
    assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ"
    assert_same_floats(gt, ot, "t")
    assert_same_floats(gu, ou, "u")
    assert_same_floats(gv, ov, "v")
    gpu_ctx.set_options(cull=1, traversal=native.TRAVERSAL_AUTO)


@pytest.mark.parametrize("trav", ["lds", "global"])
@pytest.mark.parametrize("log2_scale", [0, 52, 58])
def test_unbounded_determinants_keep_the_ieee_reciprocal(gpu_ctx, oracle, trav, log2_scale):
    """The triangle test's 1/a is a short sequence that equals the IEEE quotient for |a| <= 2^100 (tests/test_gpu_math.py), and
    the traversal drops its range test for rays whose |d|_1 x longest-edge^2 stays below 2^98 (csrc/traverse.hip, `unbounded`).
    (a) A scene scaled by 2^52 / 2^58 has edges whose squares exceed that: every ray must take the copy of the loop that keeps
    the test, and its hits (t ~ 2^52 ...) equal the oracle's plain division bit for bit. (b) Unit-size scene, directions scaled by
    2^60 ... 2^120 mixed with unit ones in one launch: same."""
    from ptmi import layout, native, scene_host, scenes
    base = scenes.random_soup(3, n_tris=500)
    k = float(2.0 ** log2_scale)
    tris = base.tris.copy()
    for f in ("v0", "v1", "v2"):
        tris[f] = (tris[f].astype(np.float64) * k).astype(np.float32)
    lights = base.lights[base.lights["light_type"] != layout.LIGHT_EMISSIVE].copy()
    nodes, depth = scene_host.build_bvh(tris)
    sc = scenes.Scene("scaled", tris, base.mats, nodes, scene_host.emissive_lights(tris, base.mats, lights), base.atlas, depth)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.set_options(cull=1, traversal=native.TRAVERSAL_GLOBAL if trav == "global" else native.TRAVERSAL_AUTO)
    o, d = _test_rays(sc, 120_000, 31)
    if log2_scale == 0:
        rng = np.random.default_rng(32)
        scale = np.exp2(rng.choice([0, 0, 60, 90, 101, 120], len(d))).astype(np.float32)
        d = (d * scale[:, None]).astype(np.float32)
    gt, gtri, gu, gv = gpu_ctx.debug_intersect(o, d)
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    assert (ot > 0).mean() > 0.1
    assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ"
    assert_same_floats(gt, ot, "t")
    assert_same_floats(gu, ou, "u")
    assert_same_floats(gv, ov, "v")
    gpu_ctx.set_options(cull=1, traversal=native.TRAVERSAL_AUTO)


@pytest.mark.parametrize("name", ["cornell", "feature_box", "cornell_spheres"])
@pytest.mark.parametrize("keep", [0, 1])
@pytest.mark.parametrize("trav", [0, 1])       # auto (LDS for these scenes) and global: there the irregular lanes read the
def test_irregular_rays_parity(gpu_ctx, oracle, scene_factory, name, keep, trav):      # uploaded tree beside lanes on the quantised one
    """Rays with zero / subnormal direction components, started exactly on box planes and triangle vertices:
    the cases where the slab test produces inf and NaN (pt.wgsl:235-236). The rebuilt hierarchy must not be
    used for them (csrc/fast_tree.hip); with keep_reference_tree=1 nothing is rebuilt at all."""
    sc = scene_factory(name)
    gpu_ctx.set_options(keep_reference_tree=keep, cull=1, traversal=trav)
    gpu_ctx.upload_scene(sc)
    rng = np.random.default_rng(21)
    n = 60_000
    corners = np.concatenate([sc.nodes["aabb_min"], sc.nodes["aabb_max"], sc.tris["v0"], sc.tris["v1"], sc.tris["v2"]])
    o = corners[rng.integers(0, len(corners), n)].astype(np.float32)
    mix = rng.integers(0, 3, n)
    o[mix == 1] += (rng.standard_normal((int((mix == 1).sum()), 3)) * 0.3).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    kind = rng.integers(0, 5, n)
    axis = rng.integers(0, 3, n)
    rows = np.arange(n)
    d[kind == 0] = 0.0
    d[rows[kind == 0], axis[kind == 0]] = rng.choice([-1.0, 1.0], int((kind == 0).sum()))       # axis-parallel
    d[rows[kind == 1], axis[kind == 1]] = 0.0                                                    # one exact zero
    d[rows[kind == 2], axis[kind == 2]] = -0.0                                                   # one negative zero
    d[rows[kind == 3], axis[kind == 3]] = np.float32(1e-41)                                      # subnormal: 1/d = inf
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30).astype(np.float32)
    d[rows[kind == 3], axis[kind == 3]] = np.float32(1e-41)
    gt, gtri, gu, gv = gpu_ctx.debug_intersect(o, d)
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    assert 0.05 < (ot > 0).mean() < 1.0
    assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ"
    assert_same_floats(gt, ot, "t")
    assert_same_floats(gu, ou, "u")
    dist = (rng.random(n) * 2.0).astype(np.float32)
    dist[::3] = -1
    assert np.array_equal(gpu_ctx.debug_occluded(o, d, dist), oracle.occluded(sc, o, d, dist))
    gpu_ctx.set_options(keep_reference_tree=0, traversal=0)
    gpu_ctx.upload_scene(sc)


def test_rebuilt_hierarchy_equals_uploaded_tree(gpu_ctx, scene_factory):
    """Same frames with the hierarchy rebuilt over the reference's leaves and with the tree exactly as uploaded,
    cull on and off: four renders, one bit pattern (the size-independent form of the equivalence argument)."""
    sc = scene_factory("cornell_spheres")
    W, H = 256, 144
    outs = []
    for keep in (0, 1):
        for cull in (1, 0):
            gpu_ctx.set_options(keep_reference_tree=keep, cull=cull, max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0,
                                frames_per_batch=0, traversal=0)
            gpu_ctx.upload_scene(sc)
            gpu_ctx.resize(W, H)
            gpu_ctx.dispatch(layout.make_camera(W, H), 8)
            outs.append(gpu_ctx.read_output())
    for o in outs[1:]:
        assert np.array_equal(bits(o), bits(outs[0]))
    gpu_ctx.set_options(keep_reference_tree=0, cull=1)


@pytest.mark.parametrize("name", ["cornell", "feature_box"])
def test_occluded_parity(gpu_ctx, oracle, scene_factory, name):
    sc = scene_factory(name)
    gpu_ctx.upload_scene(sc)
    o, d = _test_rays(sc, 200_000, 5)
    rng = np.random.default_rng(9)
    dist = (rng.random(len(o)) * 2.5).astype(np.float32)
    dist[::5] = -1.0                                        # directional-light rays
    # put some limits exactly at the closest hit distance (the t < dist - 2e-6 edge)
    ot, _, _, _, _ = oracle.intersect(sc, o, d)
    sel = (ot > 0) & (np.arange(len(o)) % 7 == 0)
    dist[sel] = ot[sel]
    sel2 = (ot > 0) & (np.arange(len(o)) % 7 == 1)
    dist[sel2] = ot[sel2] + np.float32(3e-6)
    for cull in (1, 0):
        gpu_ctx.set_options(cull=cull)
        g = gpu_ctx.debug_occluded(o, d, dist)
        r = oracle.occluded(sc, o, d, dist)
        assert 0.05 < r.mean() < 0.95
        assert np.array_equal(g, r), f"{(g != r).sum()} shadow predicates differ (cull={cull})"
    gpu_ctx.set_options(cull=1)


RENDER_CASES = [
    # scene, W, H, frames, bounces, mis, aperture
    ("cornell", 96, 64, 6, 8, 1, 0.001),
    ("cornell", 64, 64, 4, 4, 0, 0.001),          # BASELINE config 1 shape (MIS off, 4 bounces), reduced
    ("cornell_glass", 80, 60, 5, 8, 1, 0.0),
    ("feature_box", 72, 72, 6, 8, 1, 0.05),
    ("cornell_spheres", 64, 48, 3, 8, 1, 0.001),    # BASELINE configs[2] scene (atlas sampling), reduced
    ("grid_1m", 96, 54, 2, 8, 1, 0.001),            # BASELINE configs[3] scene: 999 708 triangles, depth-29 BVH
]


@pytest.mark.parametrize("case", RENDER_CASES, ids=lambda c: f"{c[0]}-{c[1]}x{c[2]}x{c[3]}-b{c[4]}-mis{c[5]}")
def test_render_parity(gpu_ctx, oracle, scene_factory, case):
    name, W, H, frames, bounces, mis, ap = case
    sc = scene_factory(name)
    cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_options(max_bounces=bounces, do_mis=mis, tile_y0=0, tile_y1=0, frames_per_batch=0, cull=1)
    gpu_ctx.reset_stats()
    gpu_ctx.dispatch(cam, frames)
    got = gpu_ctx.read_output()
    st = gpu_ctx.stats()
    assert st.segments == ost.segments, (st.segments, ost.segments)
    assert st.shadow_rays == ost.shadow_rays, (st.shadow_rays, ost.shadow_rays)
    assert st.paths == ost.paths
    assert_same_floats(got, ref, f"radiance {name}")
    assert np.isfinite(got).all() and got[..., :3].mean() > 0.01


def test_dispatch_batching_equivalence(gpu_ctx, scene_factory):
    """n_frames in one dispatch == n single-frame dispatches == any batch size (include/ptmi.h)."""
    sc = scene_factory("cornell")
    W, H, frames = 64, 48, 7
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(W, H)
    outs = []
    for fpb in (1, 3, 0):
        gpu_ctx.resize(W, H)
        gpu_ctx.set_options(max_bounces=8, do_mis=1, frames_per_batch=fpb, tile_y0=0, tile_y1=0)
        gpu_ctx.dispatch(layout.make_camera(W, H), frames)
        outs.append(gpu_ctx.read_output())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_options(frames_per_batch=0)
    for f in range(frames):                                   # the reference's loop: renderer.ts:415-454
        gpu_ctx.dispatch(layout.make_camera(W, H, frame_index=f), 1)
    outs.append(gpu_ctx.read_output())
    for o in outs[1:]:
        assert np.array_equal(bits(o), bits(outs[0]))


def test_tile_rows_and_traversal_modes(gpu_ctx, oracle, scene_factory):
    """Row bands (the multi-GPU shard unit) and every traversal mode give the same bits."""
    from ptmi import native
    sc = scene_factory("cornell")
    W, H, frames = 64, 60, 3
    cam = layout.make_camera(W, H)
    ref, _ = oracle.render(sc, cam, frames)
    gpu_ctx.upload_scene(sc)
    for mode, cull in ((native.TRAVERSAL_GLOBAL, 1), (native.TRAVERSAL_LDS, 1), (native.TRAVERSAL_GLOBAL, 0),
                       (native.TRAVERSAL_LDS, 0), (native.TRAVERSAL_GLOBAL_EXACT, 1), (native.TRAVERSAL_GLOBAL_EXACT, 0)):
        gpu_ctx.resize(W, H)
        gpu_ctx.set_options(max_bounces=8, do_mis=1, traversal=mode, cull=cull, frames_per_batch=0, tile_y0=0, tile_y1=0)
        for y0, y1 in ((0, 17), (17, 40), (40, 60)):
            gpu_ctx.set_options(tile_y0=y0, tile_y1=y1)
            gpu_ctx.dispatch(cam, frames)
        assert_same_floats(gpu_ctx.read_output(), ref, f"bands mode={mode} cull={cull}")
    gpu_ctx.set_options(traversal=native.TRAVERSAL_AUTO, cull=1, tile_y0=0, tile_y1=0)


@pytest.mark.parametrize("name,trav", [("cornell", 0), ("feature_box", 0), ("cornell_spheres", 1)])
def test_overlapped_shadow_stream_is_invisible(gpu_ctx, oracle, scene_factory, name, trav):
    """options.overlap: with 1 the shadow kernel of bounce b runs on a second stream beside the next bounce, emissive hits
    reach the radiance through records, and the record buffers alternate; without it everything is on one stream and
    `shade` adds emission itself. Both must give the oracle's bits and counters — also over several batches in one
    dispatch (frames_per_batch 2 of 5 frames: the buffers and events are reused), 1 and 2 bounces (fewer bounces than
    buffers) and with the global traversal variant (its own spill area on the side stream)."""
    sc = scene_factory(name)
    W, H, frames = 160, 100, 5
    cam = layout.make_camera(W, H)
    gpu_ctx.upload_scene(sc)
    for bounces in (8, 2, 1):
        ref, ost = oracle.render(sc, cam, frames, max_bounces=bounces, do_mis=1)
        for overlap, fpb in ((1, 0), (1, 2), (1, 5), (0, 0), (0, 2)):
            gpu_ctx.resize(W, H)
            gpu_ctx.set_options(max_bounces=bounces, do_mis=1, tile_y0=0, tile_y1=0, tile_parts=0, frames_per_batch=fpb, cull=1,
                                traversal=trav, overlap=overlap, timing=3)
            gpu_ctx.reset_stats()
            gpu_ctx.dispatch(cam, frames)
            got = gpu_ctx.read_output()
            st = gpu_ctx.stats()
            assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
            assert_same_floats(got, ref, f"radiance ({name}, overlap {overlap}, frames_per_batch {fpb}, {bounces} bounces)")
    gpu_ctx.set_options(overlap=2, frames_per_batch=0, traversal=0, max_bounces=8, timing=0)


def test_consecutive_dispatches_without_a_synchronisation(gpu_ctx, oracle, scene_factory):
    """Consecutive dispatches without a synchronisation in between (the preview loop, the benchmark's steps): the shadow stream of
    one dispatch is still running when the next one is enqueued, and the record buffers and events are reused. Six dispatches of
    4, 1, 3, 2, 6, 1 frames must leave the oracle's 17 frames."""
    sc = scene_factory("cornell")
    W, H = 128, 96
    ref, ost = oracle.render(sc, layout.make_camera(W, H), 17, max_bounces=8, do_mis=1)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, tile_parts=0, frames_per_batch=0, cull=1, traversal=0, overlap=1)
    gpu_ctx.reset_stats()
    k = 0
    for n in (4, 1, 3, 2, 6, 1):
        gpu_ctx.dispatch(layout.make_camera(W, H, frame_index=k), n)
        k += n
    got = gpu_ctx.read_output()
    st = gpu_ctx.stats()
    assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
    assert_same_floats(got, ref, "radiance after six unsynchronised dispatches")
    gpu_ctx.set_options(overlap=2)


def test_errors_are_loud(gpu_ctx, scene_factory):
    from ptmi import native
    sc = scene_factory("cornell")
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(32, 32)
    with pytest.raises(native.PtmiError):
        gpu_ctx.dispatch(layout.make_camera(64, 64), 1)            # camera / buffer size mismatch
    with pytest.raises(native.PtmiError):
        gpu_ctx.set_options(max_bounces=0)
    bad = sc.nodes.copy()
    bad["left"][0] = len(bad) + 5
    import copy
    sc2 = copy.copy(sc)
    sc2.nodes = bad
    with pytest.raises(native.PtmiError):
        gpu_ctx.upload_scene(sc2)
    gpu_ctx.upload_scene(sc)


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_fuzz(gpu_ctx, oracle, seed):
    """Seeded random scenes (scenes.random_soup): degenerate triangles, zero / flipped vertex normals (NaN shading
    normals, as the reference would compute), every material lobe, textured materials, a point light and an
    axis-aligned directional light whose shadow rays are 'irregular'. Rays, counters and every radiance bit must
    match the oracle — NaNs included — through both memory variants and with / without the rebuilt hierarchy."""
    from ptmi import native, scenes
    sc = scenes.random_soup(seed)
    W, H, frames = 64, 48, 4
    cam = layout.make_camera(W, H, aperture=0.02 if seed % 2 else 0.0, focus_distance=2.5)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    rng = np.random.default_rng(100 + seed)
    n = 4096
    o = (rng.random((n, 3)) * [2.4, 2.4, 2.4] + [-1.2, -0.2, -1.2]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::17, rng.integers(0, 3)] = 0.0                                  # some irregular rays
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t_ref, tri_ref, u_ref, v_ref, _ = oracle.intersect(sc, o, d)
    for keep, trav in ((0, native.TRAVERSAL_AUTO), (1, native.TRAVERSAL_GLOBAL), (0, native.TRAVERSAL_GLOBAL),
                       (0, native.TRAVERSAL_GLOBAL_EXACT)):
        gpu_ctx.set_options(keep_reference_tree=keep)
        gpu_ctx.upload_scene(sc)
        gpu_ctx.resize(W, H)
        gpu_ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, frames_per_batch=0, cull=1, traversal=trav)
        t, tri, u, v = gpu_ctx.debug_intersect(o, d)
        assert np.array_equal(tri, tri_ref)
        assert_same_floats(t, t_ref, "t"); assert_same_floats(u, u_ref, "u"); assert_same_floats(v, v_ref, "v")
        gpu_ctx.reset_stats()
        gpu_ctx.dispatch(cam, frames)
        got = gpu_ctx.read_output()
        st = gpu_ctx.stats()
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance (seed {seed}, keep {keep}, traversal {trav})")
    gpu_ctx.set_options(keep_reference_tree=0, traversal=native.TRAVERSAL_AUTO)
    assert np.nanmean(ref[..., :3]) > 0.005


@pytest.mark.parametrize("parts,strip,rows", [(3, 4, (0, 0)), (2, 1, (0, 0)), (4, 5, (7, 58)), (8, 4, (0, 0)), (70, 1, (0, 0))])
def test_interleaved_strips_cover_the_frame(gpu_ctx, oracle, scene_factory, parts, strip, rows):
    """tile_parts / tile_part / tile_strip (include/ptmi.h): N contexts with tile_part = 0..N-1 render disjoint
    strips that together are the single render, bit for bit — also with a short last strip, a sub-range of rows and
    more parts than strips."""
    sc = scene_factory("cornell")
    W, H, frames = 64, 60, 3
    cam = layout.make_camera(W, H)
    ref, ost = oracle.render(sc, cam, frames)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_options(max_bounces=8, do_mis=1, frames_per_batch=0, tile_y0=rows[0], tile_y1=rows[1], tile_parts=parts,
                        tile_strip=strip)
    gpu_ctx.reset_stats()
    for part in range(parts):
        gpu_ctx.set_options(tile_part=part)
        gpu_ctx.dispatch(cam, frames)
    got = gpu_ctx.read_output()
    y0, y1 = rows[0], rows[1] or H
    assert_same_floats(got[y0:y1], ref[y0:y1], f"strips parts={parts} strip={strip}")
    assert not got[:y0].any() and not got[y1:].any()
    if rows == (0, 0):
        assert gpu_ctx.stats().segments == ost.segments
    with pytest.raises(Exception):
        gpu_ctx.set_options(tile_parts=2, tile_part=2)
    gpu_ctx.set_options(tile_y0=0, tile_y1=0, tile_parts=0, tile_part=0, tile_strip=0)
