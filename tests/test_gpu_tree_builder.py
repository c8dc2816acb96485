"""ptmi_options.tree_builder = 2: the traversal hierarchy over the uploaded leaves is built on the GPU (csrc/gpu_tree.hip: Morton-order
linear BVH — radix sort, Karras' radix tree, bottom-up fit) instead of by the host's SAH builder. The leaves (triangle ranges, exact boxes)
are the reference's either way and inner boxes are exact unions, so every (t, triangle, u, v), every shadow predicate and every radiance
bit must equal the oracle's — through the LDS kernels, the quantised image and the exact image in global memory, with rays of every kind.
Reference: src/renderer/bvh.ts:53-157 builds the leaves; src/shader/pt.wgsl:248-291 walks them."""
import numpy as np
import pytest

from ptmi import layout
from test_gpu_parity import _test_rays, assert_same_floats

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tb_ctx(gpu_ctx):
    from ptmi import native
    before = gpu_ctx.options().leaves
    gpu_ctx.set_options(tree_builder=2, keep_reference_tree=0, leaves=1)        # the builder of the hierarchy over the REFERENCE's leaves
    yield gpu_ctx
    gpu_ctx.set_options(leaves=before, tree_builder=0, traversal=native.TRAVERSAL_AUTO, cull=1, keep_reference_tree=0, overlap=2, frames_per_batch=0,
                        max_bounces=8, do_mis=1)


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "feature_box", "grid_1m"])
def test_gpu_built_tree_extend_and_shadow_parity(tb_ctx, oracle, scene_factory, name):
    from ptmi import native
    sc = scene_factory(name)
    tb_ctx.upload_scene(sc)
    n = 120_000 if name == "grid_1m" else 200_000
    o, d = _test_rays(sc, n, 33)
    d[::23, 1] = 0.0                                           # irregular rays walk the uploaded tree beside the others
    d[::31] *= np.float32(3.0)
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    rng = np.random.default_rng(5)
    dist = (rng.random(len(o)) * 2.5).astype(np.float32)
    dist[::5] = -1.0
    occ_ref = oracle.occluded(sc, o, d, dist)
    for trav in (native.TRAVERSAL_AUTO, native.TRAVERSAL_GLOBAL, native.TRAVERSAL_GLOBAL_EXACT):
        for cull in (1, 0):
            tb_ctx.set_options(traversal=trav, cull=cull)
            gt, gtri, gu, gv = tb_ctx.debug_intersect(o, d)
            assert np.array_equal(gtri, otri), f"{(gtri != otri).sum()} triangle ids differ (traversal {trav}, cull {cull})"
            assert_same_floats(gt, ot, "t"); assert_same_floats(gu, ou, "u"); assert_same_floats(gv, ov, "v")
            assert np.array_equal(tb_ctx.debug_occluded(o, d, dist), occ_ref)


@pytest.mark.parametrize("name,W,H,frames", [("cornell", 160, 100, 4), ("cornell_spheres", 96, 64, 3), ("feature_box", 72, 72, 4)])
def test_gpu_built_tree_render_parity(tb_ctx, oracle, scene_factory, name, W, H, frames):
    sc = scene_factory(name)
    cam = layout.make_camera(W, H, aperture=0.01, focus_distance=2.8)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    tb_ctx.upload_scene(sc)
    tb_ctx.resize(W, H)
    tb_ctx.reset_stats()
    tb_ctx.dispatch(cam, frames)
    got = tb_ctx.read_output()
    st = tb_ctx.stats()
    assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
    assert_same_floats(got, ref, f"radiance ({name}, hierarchy built on the GPU)")


@pytest.mark.parametrize("seed", range(4))
def test_gpu_built_tree_random_scene_fuzz(tb_ctx, oracle, seed):
    from ptmi import scenes
    sc = scenes.random_soup(80 + seed)                      # degenerate triangles, coincident centroids: equal Morton codes
    W, H, frames = 64, 48, 3
    cam = layout.make_camera(W, H, aperture=0.02 if seed % 2 else 0.0, focus_distance=2.5)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    tb_ctx.upload_scene(sc)
    tb_ctx.resize(W, H)
    tb_ctx.reset_stats()
    tb_ctx.dispatch(cam, frames)
    st = tb_ctx.stats()
    assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
    assert_same_floats(tb_ctx.read_output(), ref, f"radiance (seed {seed})")
