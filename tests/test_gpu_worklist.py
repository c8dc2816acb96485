"""GPU parity of the per-wave work list (ptmi_options.worklist = 2, csrc/traverse.hip trace_wave_wl): lanes list the triangles of
the leaves they open as (ray lane, triangle) items in LDS, all 64 lanes take one item each, the closest hit is an LDS 64-bit
minimum on (bits(t), triangle). Other lanes test the triangles, in another order — every (t, triangle, u, v), every shadow
predicate, every counter and every radiance bit must equal the oracle's (and so the per-lane loop's, which the rest of the
GPU suite pins). Reference: src/shader/pt.wgsl:248-296 (traverseBVH), :274 (strictly nearer wins, first found on ties)."""
import numpy as np
import pytest

from ptmi import layout
from test_gpu_parity import _test_rays, assert_same_floats

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wl_ctx(gpu_ctx):
    from ptmi import native
    gpu_ctx.set_options(worklist=2, traversal=native.TRAVERSAL_LDS, cull=1, keep_reference_tree=0)
    yield gpu_ctx
    gpu_ctx.set_options(worklist=0, traversal=native.TRAVERSAL_AUTO, cull=1, keep_reference_tree=0, overlap=2, frames_per_batch=0,
                        max_bounces=8, do_mis=1)


def _big_leaf_scene(seed, max_leaf, n_tris=600):
    """random_soup's triangles under a BVH whose leaves hold up to max_leaf triangles (the reference builds 4; the ABI takes 32):
    counts beyond 3 bits, and leaves that do not fit what is left of a ring."""
    from ptmi import scene_host, scenes
    sc = scenes.random_soup(seed, n_tris=n_tris)
    tris = sc.tris.copy()
    nodes, depth = scene_host.build_bvh(tris, max_leaf=max_leaf)
    lights = scene_host.emissive_lights(tris, sc.mats, sc.lights[sc.lights["light_type"] != layout.LIGHT_EMISSIVE])
    return scenes.Scene(f"soup{seed}_leaf{max_leaf}", tris, sc.mats, nodes, lights, sc.atlas, depth, {})


@pytest.mark.parametrize("name", ["cornell", "feature_box", "cornell_glass"])
@pytest.mark.parametrize("cull", [1, 0])
def test_worklist_extend_parity(wl_ctx, oracle, scene_factory, name, cull):
    sc = scene_factory(name)
    wl_ctx.upload_scene(sc)
    wl_ctx.set_options(cull=cull)
    o, d = _test_rays(sc, 300_000, 21)
    d[::23, 1] = 0.0                                           # irregular rays walk the uploaded tree beside the others
    d[::31] *= np.float32(3.0)                                 # un-normalised directions
    gt, gtri, gu, gv = wl_ctx.debug_intersect(o, d)
    assert wl_ctx.stats().worklist_used & 1, "the work-list kernel did not run"
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    assert (ot > 0).mean() > 0.4
    assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ"
    assert_same_floats(gt, ot, "t"); assert_same_floats(gu, ou, "u"); assert_same_floats(gv, ov, "v")


@pytest.mark.parametrize("name", ["cornell", "feature_box"])
def test_worklist_occluded_parity(wl_ctx, oracle, scene_factory, name):
    sc = scene_factory(name)
    wl_ctx.upload_scene(sc)
    o, d = _test_rays(sc, 200_000, 15)
    rng = np.random.default_rng(19)
    dist = (rng.random(len(o)) * 2.5).astype(np.float32)
    dist[::5] = -1.0
    ot, _, _, _, _ = oracle.intersect(sc, o, d)
    sel = (ot > 0) & (np.arange(len(o)) % 7 == 0)
    dist[sel] = ot[sel]                                        # the t < dist - 2e-6 edge of pt.wgsl:423/465
    sel2 = (ot > 0) & (np.arange(len(o)) % 7 == 1)
    dist[sel2] = ot[sel2] + np.float32(3e-6)
    for cull in (1, 0):
        wl_ctx.set_options(cull=cull)
        g = wl_ctx.debug_occluded(o, d, dist)
        assert wl_ctx.stats().worklist_used & 2, "the work-list kernel did not run"
        r = oracle.occluded(sc, o, d, dist)
        assert np.array_equal(g, r), f"{(g != r).sum()} shadow predicates differ (cull={cull})"


@pytest.mark.parametrize("max_leaf", [1, 7, 32])
def test_worklist_with_other_leaf_sizes(wl_ctx, oracle, max_leaf):
    sc = _big_leaf_scene(3, max_leaf, 200 if max_leaf == 1 else 600)            # (one-triangle leaves: a deeper tree per triangle)
    assert sc.nodes["triangle_count"].max() <= max_leaf and (max_leaf == 1 or sc.nodes["triangle_count"].max() > 4)
    wl_ctx.upload_scene(sc)
    rng = np.random.default_rng(5)
    n = 100_000
    o = (rng.random((n, 3)) * [2.4, 2.4, 2.4] + [-1.2, -0.2, -1.2]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gt, gtri, gu, gv = wl_ctx.debug_intersect(o, d)
    assert wl_ctx.stats().worklist_used & 1, "the work-list kernel did not run"
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ"
    assert_same_floats(gt, ot, "t"); assert_same_floats(gu, ou, "u"); assert_same_floats(gv, ov, "v")
    W, H, frames = 64, 48, 3
    cam = layout.make_camera(W, H, aperture=0.0)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    wl_ctx.resize(W, H)
    wl_ctx.reset_stats()
    wl_ctx.dispatch(cam, frames)
    got = wl_ctx.read_output()
    st = wl_ctx.stats()
    assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
    assert_same_floats(got, ref, f"radiance (leaves of up to {max_leaf})")


@pytest.mark.parametrize("name,W,H,frames,bounces,mis,ap", [
    ("cornell", 200, 130, 5, 8, 1, 0.001), ("cornell", 64, 64, 4, 4, 0, 0.001), ("cornell_glass", 80, 60, 5, 8, 1, 0.0),
    ("feature_box", 72, 72, 6, 8, 1, 0.05)])
def test_worklist_render_parity(wl_ctx, oracle, scene_factory, name, W, H, frames, bounces, mis, ap):
    sc = scene_factory(name)
    cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
    wl_ctx.upload_scene(sc)
    for overlap, fpb in ((1, 0), (0, 0), (1, 2)):
        wl_ctx.resize(W, H)
        wl_ctx.set_options(max_bounces=bounces, do_mis=mis, tile_y0=0, tile_y1=0, tile_parts=0, frames_per_batch=fpb, overlap=overlap)
        wl_ctx.reset_stats()
        wl_ctx.dispatch(cam, frames)
        got = wl_ctx.read_output()
        st = wl_ctx.stats()
        assert st.worklist_used == (3 if mis else 1), "the work-list kernels did not run"
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance ({name}, overlap {overlap}, frames_per_batch {fpb})")


@pytest.mark.parametrize("seed", range(4))
def test_worklist_random_scene_fuzz(wl_ctx, oracle, seed):
    from ptmi import scenes
    sc = scenes.random_soup(40 + seed)
    W, H, frames = 64, 48, 4
    cam = layout.make_camera(W, H, aperture=0.02 if seed % 2 else 0.0, focus_distance=2.5)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    for keep in (0, 1):
        wl_ctx.set_options(keep_reference_tree=keep)
        wl_ctx.upload_scene(sc)
        wl_ctx.resize(W, H)
        wl_ctx.reset_stats()
        wl_ctx.dispatch(cam, frames)
        got = wl_ctx.read_output()
        st = wl_ctx.stats()
        assert st.worklist_used == 3, "the work-list kernels did not run"
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance (seed {seed}, keep {keep})")
