"""Edge cases of the boundary on the GPU, each against the oracle bit for bit: empty and one-leaf scenes,
1x1 and ragged image sizes, widths above 1000 (where the reference's seed x + 1000 y + 100000 frame aliases
pixels), one bounce, zero frames, frame indices far from zero, lights-free scenes, resume from a saved buffer."""
import numpy as np
import pytest

from ptmi import layout, scene_host, scenes

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def render_both(gpu_ctx, oracle, sc, cam, frames, **opt):
    o = dict(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, frames_per_batch=0, cull=1, traversal=0)
    o.update(opt)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(int(cam["width"]), int(cam["height"]))
    gpu_ctx.set_options(**o)
    gpu_ctx.reset_stats()
    gpu_ctx.dispatch(cam, frames)
    got, st = gpu_ctx.read_output(), gpu_ctx.stats()
    ref, ost = oracle.render(sc, cam, frames, max_bounces=o["max_bounces"], do_mis=o["do_mis"])
    assert (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays)
    assert same(got, ref)
    return got, st


def tiny_scene(n_tris):
    """n_tris triangles of the Cornell floor/light (0 -> empty blobs; <= 4 -> the root is a leaf)."""
    full = scenes.make("cornell")
    mats = full.mats
    if n_tris == 0:
        return scenes.Scene("empty", np.zeros(0, layout.TRIANGLE), mats, np.zeros(0, layout.BVH_NODE),
                            np.zeros(0, layout.LIGHT), None)
    idx = np.r_[np.flatnonzero(full.tris["material_index"] == 3), np.flatnonzero(full.tris["material_index"] == 0)][:n_tris]
    tris = np.ascontiguousarray(full.tris[idx])
    nodes, depth = scene_host.build_bvh(tris)
    return scenes.Scene(f"tiny{n_tris}", tris, mats, nodes, scene_host.emissive_lights(tris, mats), None, depth)


def test_empty_scene_is_black(gpu_ctx, oracle):
    got, st = render_both(gpu_ctx, oracle, tiny_scene(0), layout.make_camera(33, 17), 3)
    assert not got.any() and st.segments == 33 * 17 * 3 and st.shadow_rays == 0


@pytest.mark.parametrize("n", [1, 2, 4, 5])
def test_root_leaf_and_smallest_trees(gpu_ctx, oracle, n):
    sc = tiny_scene(n)
    assert (len(sc.nodes) == 1) == (n <= 4)
    got, _ = render_both(gpu_ctx, oracle, sc, layout.make_camera(40, 30, position=(0, 1.0, 1.5), aperture=0.0), 4)
    assert got[..., :3].max() > 0          # the light quad is in view


@pytest.mark.parametrize("W,H", [(1, 1), (1, 7), (7, 1), (63, 5), (65, 3), (1030, 4)])
def test_ragged_sizes_and_seed_aliasing(gpu_ctx, oracle, scene_factory, W, H):
    # 1030 wide: pixels (x + 1000, y) and (x, y + 1) share a seed (random.wgsl:3-5) — kept, not fixed
    render_both(gpu_ctx, oracle, scene_factory("cornell"), layout.make_camera(W, H), 3)


def test_one_bounce_no_mis_and_high_frame_index(gpu_ctx, oracle, scene_factory):
    sc = scene_factory("cornell")
    render_both(gpu_ctx, oracle, sc, layout.make_camera(48, 32), 2, max_bounces=1, do_mis=0)
    render_both(gpu_ctx, oracle, sc, layout.make_camera(48, 32), 2, max_bounces=1, do_mis=1)
    # frame index near the u32 wrap of frame * 100000 (42950 * 100000 > 2^32)
    cam = layout.make_camera(32, 24, frame_index=42949)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(32, 24)
    gpu_ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0)
    prev = np.random.default_rng(1).random((24, 32, 4)).astype(np.float32)
    prev[..., 3] = 0
    gpu_ctx.write_output(prev)
    gpu_ctx.dispatch(cam, 3)
    ref, _ = oracle.render(sc, cam, 3, out=prev.copy())
    assert same(gpu_ctx.read_output(), ref)


def test_zero_frames_and_resume(gpu_ctx, oracle, scene_factory):
    sc = scene_factory("feature_box")
    W, H = 40, 28
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, frames_per_batch=0)
    gpu_ctx.dispatch(layout.make_camera(W, H), 0)                       # a no-op
    assert not gpu_ctx.read_output().any()
    gpu_ctx.dispatch(layout.make_camera(W, H), 5)
    saved = gpu_ctx.read_output()
    gpu_ctx.resize(W, H)                                                # drops the buffer (renderer.ts:496-510)
    assert not gpu_ctx.read_output().any()
    gpu_ctx.write_output(saved)                                         # resume = buffer + frame index
    gpu_ctx.dispatch(layout.make_camera(W, H, frame_index=5), 4)
    ref, _ = oracle.render(sc, layout.make_camera(W, H), 9)
    assert same(gpu_ctx.read_output(), ref)


def test_scene_without_lights(gpu_ctx, oracle):
    sc = scenes.make("cornell")
    import copy
    dark = copy.copy(sc)
    dark.lights = np.zeros(0, layout.LIGHT)                             # emissive surfaces stay, NEE has nothing to pick
    got, st = render_both(gpu_ctx, oracle, dark, layout.make_camera(48, 36), 4)
    assert st.shadow_rays == 0 and got[..., :3].max() > 0


def test_upload_validation_is_loud(gpu_ctx, scene_factory):
    from ptmi import native
    import copy
    sc = scene_factory("cornell")
    for mutate in (lambda s: s.nodes.__setitem__(0, s.nodes[5]),                       # root becomes a copy: a cycle-free but wrong tree
                   lambda s: s.lights["triangle_index"].__setitem__(0, 10 ** 6),       # light -> missing triangle
                   lambda s: s.lights["light_type"].__setitem__(0, 9)):                # unknown light type
        bad = copy.copy(sc)
        bad.nodes, bad.lights = sc.nodes.copy(), sc.lights.copy()
        mutate(bad)
        try:
            gpu_ctx.upload_scene(bad)
        except native.PtmiError:
            continue
        # a structurally valid (if different) tree may be accepted; it must then still render without faulting
        gpu_ctx.resize(16, 16)
        gpu_ctx.dispatch(layout.make_camera(16, 16), 1)
        gpu_ctx.read_output()
    leafy = copy.copy(sc)
    leafy.nodes = sc.nodes.copy()
    leaf = int(np.flatnonzero(leafy.nodes["triangle_count"] > 0)[0])
    leafy.nodes["triangle_count"][leaf] = 40                                            # > 32 triangles in a leaf
    with pytest.raises(native.PtmiError):
        gpu_ctx.upload_scene(leafy)
    gpu_ctx.upload_scene(sc)


def test_deep_tree_spills_the_node_stack(gpu_ctx, oracle, scene_factory):
    """scenes.deep_chain: a 35-level BVH. The global variant keeps 16 stack entries per lane in LDS and moves the rest
    of a deep node stack to its spill area; rays along the chain (one pending far child per level) force that, with
    and without the rebuilt hierarchy, closest hit and any hit — results must stay those of the oracle."""
    from ptmi import native
    sc = scene_factory("deep_chain")
    assert sc.bvh_depth >= 33
    rng = np.random.default_rng(9)
    n = 60_000
    c = (sc.tris["v0"].astype(np.float64) + sc.tris["v1"] + sc.tris["v2"]) / 3
    scale = np.linalg.norm(sc.tris["v1"].astype(np.float64) - sc.tris["v0"], axis=1)
    pick = rng.integers(0, len(c), n)
    # a third: rays along the chain from before its start; a third: aimed at a triangle from nearby; a third: random
    o = np.zeros((n, 3)); d = np.zeros((n, 3))
    k = n // 3
    o[:k] = [-1e-12, 0, 0] + rng.normal(size=(k, 3)) * 1e-13; d[:k] = [1, 0, 0] + rng.normal(size=(k, 3)) * rng.choice([0, 1e-3, 0.05], (k, 1))
    o[k:2 * k] = c[pick[k:2 * k]] + rng.normal(size=(k, 3)) * scale[pick[k:2 * k], None] * 3
    d[k:2 * k] = c[pick[k:2 * k]] + rng.normal(size=(k, 3)) * scale[pick[k:2 * k], None] * 0.3 - o[k:2 * k]
    o[2 * k:] = c[pick[2 * k:]] * rng.random((n - 2 * k, 1)) * 2 + rng.normal(size=(n - 2 * k, 3)) * scale[pick[2 * k:], None]
    d[2 * k:] = rng.normal(size=(n - 2 * k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o, d = o.astype(np.float32), d.astype(np.float32)
    t_ref, tri_ref, u_ref, v_ref, _ = oracle.intersect(sc, o, d)
    assert (t_ref > 0).mean() > 0.15
    dist = np.where(rng.random(n) < 0.5, -1.0, np.abs(rng.normal(size=n)) * scale[pick] * 4).astype(np.float32)
    occ_ref = oracle.occluded(sc, o, d, dist)
    for keep in (0, 1):
        for trav in (native.TRAVERSAL_GLOBAL, native.TRAVERSAL_AUTO):
            gpu_ctx.set_options(keep_reference_tree=keep)
            gpu_ctx.upload_scene(sc)
            gpu_ctx.set_options(traversal=trav, cull=1)
            t, tri, u, v = gpu_ctx.debug_intersect(o, d)
            assert np.array_equal(tri, tri_ref), (keep, trav, int((tri != tri_ref).sum()))
            assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32)) and np.array_equal(u.view(np.uint32), u_ref.view(np.uint32))
            assert np.array_equal(gpu_ctx.debug_occluded(o, d, dist), occ_ref)
    gpu_ctx.set_options(keep_reference_tree=0, traversal=native.TRAVERSAL_AUTO)


def test_distance_cull_against_grazing_triangles(gpu_ctx, oracle):
    """The one freedom of the traversal that is not exact arithmetic: cull = 1 skips a box whose
    entry distance exceeds best * 1.001 + 1e-4 — slack for the rounding of a triangle's own t. That slack covers every
    triangle whose computed t is within 0.1 % of its true distance; Moller-Trumbore (pt.wgsl:128-158) loses that accuracy
    only for rays within ~1e-4 rad of a large triangle's plane, where t = (s.N)/(d.N) divides two cancelling sums.
    This scene builds exactly that: a blocker 50 units away and, behind it, a 20-unit triangle pair whose plane the rays
    graze (the oracle takes the pair as the nearest hit on ~100 of the 20 000 grazing rays, at computed distances below 49.9
    for a surface that starts at 50.5; on ~40 of them the pair's leaf box lies beyond the cull limit once the blocker is
    found). Checked:
      * cull = 0 (the reference's own leaf set) equals the oracle bit for bit on every ray, grazing or not;
      * cull = 1 equals it too on every ray that meets the planes at more than 1e-2 rad;
      * where cull = 1 differs on a grazing ray, the oracle's pick is the grazing triangle with a computed t that is off
        its true distance by more than the slack — the documented caveat, nothing else."""
    def grid(corner, du, dv, nu, nv, normal):
        c, du, dv = (np.array(x, np.float64) for x in (corner, du, dv))
        return [scenes._quad(c + du * i + dv * j, c + du * (i + 1) + dv * j, c + du * (i + 1) + dv * (j + 1), c + du * i + dv * (j + 1),
                             normal, 0) for i in range(nu) for j in range(nv)]
    parts = grid((0, -1, -1), (0, 0.5, 0), (0, 0, 0.5), 4, 4, (-1, 0, 0))                    # blocker: 32 triangles in the plane x = 0
    # the grazed pair: a 20 x 2 strip, so that (even rotated) its box starts behind the blocker for rays along it
    parts.append(scenes._quad((0.5, 0.3, -1), (20.5, 0.3, -1), (20.5, 0.3, 1), (0.5, 0.3, 1), (0, -1, 0), 0))
    parts += grid((30, -20, -20), (0, 4, 0), (0, 0, 4), 10, 10, (-1, 0, 0))                  # far wall: 200 triangles
    # nothing axis-aligned: with axis-aligned triangles most products in Moller-Trumbore are exact zeros and nothing cancels
    def rot(ax, a):
        c, s_ = np.cos(a), np.sin(a)
        return np.array([[[1, 0, 0], [0, c, -s_], [0, s_, c]], [[c, 0, s_], [0, 1, 0], [-s_, 0, c]], [[c, -s_, 0], [s_, c, 0], [0, 0, 1]]][ax])
    R = rot(2, 0.5) @ rot(0, 0.35) @ rot(1, 0.2)
    for t in parts:
        for k in ("v0", "v1", "v2", "n0", "n1", "n2"):
            t[k] = (t[k].astype(np.float64) @ R.T).astype(np.float32)
    sc = scenes._finish("grazing", parts, [scenes._material()])
    big_ids = np.flatnonzero(np.isclose(sc.tris["v0"].astype(np.float64) @ R, 0.3, atol=1e-5)[:, 1]
                             & np.isclose(sc.tris["v1"].astype(np.float64) @ R, 0.3, atol=1e-5)[:, 1]
                             & np.isclose(sc.tris["v2"].astype(np.float64) @ R, 0.3, atol=1e-5)[:, 1])
    assert len(big_ids) == 2
    rng = np.random.default_rng(5)
    n = 40_000
    theta = np.where(np.arange(n) % 2 == 0, 10.0 ** rng.uniform(-7.5, -4, n), 10.0 ** rng.uniform(-2, -0.6, n))   # grazing / control
    xh = rng.uniform(1.0, 20.0, n)                      # where the ray would meet the grazed plane (behind the blocker)
    z0 = rng.uniform(-0.9, 0.9, n)
    d = np.stack([np.ones(n), theta, np.zeros(n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.stack([np.full(n, -50.0), 0.3 - theta * (50.0 + xh), z0], 1)
    o, d = (o @ R.T).astype(np.float32), (d @ R.T).astype(np.float32)
    gpu_ctx.upload_scene(sc)
    ot, otri, ou, ov, _ = oracle.intersect(sc, o, d)
    res = {}
    for cull in (0, 1):
        gpu_ctx.set_options(cull=cull, traversal=0)
        res[cull] = gpu_ctx.debug_intersect(o, d)
    gpu_ctx.set_options(cull=1)
    t0, tri0, u0, v0 = res[0]
    assert np.array_equal(tri0, otri) and same(t0, ot) and same(u0, ou) and same(v0, ov)          # cull = 0: exact everywhere
    t1, tri1, u1, v1 = res[1]
    differ = (tri1 != otri) | (t1.view(np.uint32) != ot.view(np.uint32))
    control = theta >= 1e-2
    assert control.sum() > 10_000 and not differ[control].any()                                   # cull = 1: exact off the grazing band
    # true distance to the grazed plane, in float64 from the very float32 inputs the kernels saw
    T = sc.tris[big_ids[0]]
    p0 = T["v0"].astype(np.float64)
    N = np.cross(T["v1"].astype(np.float64) - p0, T["v2"].astype(np.float64) - p0)
    true_t = ((p0 - o.astype(np.float64)) @ N) / (d.astype(np.float64) @ N)
    for i in np.flatnonzero(differ):
        assert otri[i] in big_ids                                                                 # the oracle picked a grazing triangle
        assert abs(ot[i] - true_t[i]) > 1e-3 * abs(true_t[i])                                     # ... at a distance it is not at
    print(f"grazing rays: {int((~control).sum())}, cull=1 differs from the reference's result on {int(differ.sum())}")
