"""Pins for the CPU oracle (oracle/pt_oracle.c).

The reference holds no golden vectors for the path-tracing path, so the
oracle is pinned by: the integer RNG known answers of SURVEY.md Appendix B (re-derived below
with Python integers, independently of the oracle's code), analytic intersection cases,
closed-form BSDF / sampling identities, an emission furnace with a closed-form answer, and
agreement between the contract build and the literal (PT_STRICT) build.
"""
import math

import numpy as np
import pytest

from ptmi import layout, scene_host, scenes

M32 = 0xFFFFFFFF


def py_rand(state):
    """random.wgsl:7-12 in Python integers."""
    state = (state * 747796405 + 2891336453) & M32
    w = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & M32
    w = (w >> 22) ^ w
    return state, w


# SURVEY.md Appendix B (x, y, frame) -> seed, then (state, word) of draws 1..4
APPENDIX_B = {
    (0, 0, 0): (0, [(2891336453, 0x07BB2FE2), (1192405134, 0x22B6B6BC), (568162667, 0x3BF6E0B1), (878960812, 0x572F7439)]),
    (1, 0, 0): (1, [(3639132858, 0xA8BEEA3C), (1098935943, 0x2679C518), (3968020856, 0x97AAF6C6), (3079081181, 0x2A521372)]),
    (3, 2, 5): (502003, [(1608496084, 0x68F850A6), (1895824361, 0x4ADB238D), (1179563458, 0xC755A6C2), (4179491631, 0xAA91A486)]),
    (1919, 1079, 63): (7380919, [(3767748712, 0xC74A0B71), (376493197, 0x11F96686), (1901338038, 0x4D0EE4D9), (543876787, 0xB3232B0D)]),
    (1000, 0, 0): (1000, [(3363431949, 0x80FA0A57), (2212494646, 0x359E6CB7), (3107220531, 0xBB32C3CF), (1976074260, 0x17241C38)]),
}
APPENDIX_B_FLOATS = {(0, 0, 0): [0.0301999971, 0.135600492, 0.234235808, 0.340567827],
                     (3, 2, 5): [0.410038978, 0.292406291, 0.778650701, 0.666284859]}


@pytest.mark.parametrize("key", list(APPENDIX_B))
def test_rng_known_answers(oracle, key):
    seed, draws = APPENDIX_B[key]
    assert oracle.seed(*key) == seed == (key[0] + 1000 * key[1] + 100000 * key[2]) & M32
    states, words, vals = oracle.rand(seed, 4)
    st = seed
    for i, (s_ref, w_ref) in enumerate(draws):
        st, w = py_rand(st)
        assert (st, w) == (s_ref, w_ref)                       # table == integer re-derivation
        assert (int(states[i]), int(words[i])) == (s_ref, w_ref)
        assert vals[i] == np.float32(w_ref) / np.float32(4294967296.0)
    if key in APPENDIX_B_FLOATS:
        assert np.allclose(vals, APPENDIX_B_FLOATS[key], rtol=0, atol=5e-9)


def test_rng_seed_collisions_and_unit_interval(oracle):
    assert oracle.seed(1000, 0, 0) == oracle.seed(0, 1, 0)      # W > 1000 aliases (x+1000, y) = (x, y+1)
    assert oracle.seed(5, 107, 3) == oracle.seed(5, 7, 4)        # (x, y+100, f) = (x, y, f+1)
    assert oracle.seed(0, 0, 42950) == (42950 * 100000) & M32    # u32 wrap
    _, words, vals = oracle.rand(12345, 200000)
    assert vals.min() >= 0.0 and vals.max() <= 1.0
    assert np.array_equal(vals, words.astype(np.float32) / np.float32(4294967296.0))
    # f32(word) rounds to 2^32 for word >= 2^32 - 128: rand() can return exactly 1.0
    assert np.float32(np.uint32(0xFFFFFF80)) / np.float32(4294967296.0) == np.float32(1.0)
    k, _ = oracle.rand_int(777, 0, 1)
    assert k in (0, 1)
    ks = [oracle.rand_int(s, 0, 6)[0] for s in range(2000)]
    assert min(ks) == 0 and max(ks) == 6


def test_sincos_contract(oracle):
    xs = np.linspace(0, 2 * math.pi, 20001).astype(np.float32)
    err = 0.0
    for x in xs[::7]:
        s, c = oracle.sincos(float(x))
        err = max(err, abs(s - math.sin(float(x))), abs(c - math.cos(float(x))))
    assert err < 2e-7
    assert oracle.sincos(0.0) == (0.0, 1.0)


# ---- analytic intersection ----------------------------------------------------------------
def one_triangle_scene(extra=None):
    t = np.zeros(1 if extra is None else 2, layout.TRIANGLE)
    t["v0"][0], t["v1"][0], t["v2"][0] = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    for k in ("n0", "n1", "n2"):
        t[k] = (0, 0, 1)
    if extra is not None:
        t["v0"][1], t["v1"][1], t["v2"][1] = extra
    mats = np.zeros(1, layout.MATERIAL)
    mats["base_color"] = 0.8
    mats["roughness"] = 0.5
    mats["ior"] = 1.5
    nodes, depth = scene_host.build_bvh(t)
    return scenes.Scene("one", t, mats, nodes, np.zeros(0, layout.LIGHT), None, depth)


def test_ray_triangle_analytic(oracle):
    sc = one_triangle_scene()
    o = np.array([[0.25, 0.25, 1.0], [0.25, 0.25, -2.0], [2.0, 2.0, 1.0], [0.25, 0.25, 1.0], [0.25, 0.25, 1.0],
                  [0.25, 0.25, 5e-7], [0.5, 0.5, 3.0], [0.0, 0.0, 1.0]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [1, 0, 0],
                  [0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    t, tri, u, v, _ = oracle.intersect(sc, o, d)
    assert t[0] == 1.0 and tri[0] == 0 and u[0] == 0.25 and v[0] == 0.25       # front hit: exact in f32
    assert t[1] == 2.0 and tri[1] == 0                                          # back face is hit too (pt.wgsl:134 uses abs)
    assert t[2] == -1.0 and tri[2] == 0xFFFFFFFF                                # outside the triangle
    assert t[3] == -1.0                                                         # pointing away: t < 0
    assert t[4] == -1.0                                                         # parallel: |a| < 1e-6
    assert t[5] == -1.0                                                         # t = 5e-7 <= EPSILON is rejected (pt.wgsl:157)
    assert t[6] == 3.0 and u[6] == 0.5 and v[6] == 0.5                          # on the hypotenuse: u+v = 1 is inside
    # origin on a slab plane with a zero direction component: (min - o)/d = 0/0 = NaN in the slab test
    # (pt.wgsl:235); WGSL leaves min/max of NaN open, the contract takes the other operand, which
    # turns that axis into [inf, inf] and rejects the box — the triangle is never tested
    assert t[7] == -1.0


def test_closest_and_tie_break(oracle):
    # two coplanar duplicate triangles: equal t -> the first in DFS-left order = lowest index (pt.wgsl:275)
    sc = one_triangle_scene(extra=((0, 0, 0), (1, 0, 0), (0, 1, 0)))
    t, tri, _, _, _ = oracle.intersect(sc, [[0.2, 0.2, 1.0]], [[0, 0, -1]])
    assert t[0] == 1.0 and tri[0] == 0
    # a nearer triangle with the higher index wins on distance
    sc = one_triangle_scene(extra=((0, 0, 0.5), (1, 0, 0.5), (0, 1, 0.5)))
    t, tri, _, _, _ = oracle.intersect(sc, [[0.2, 0.2, 1.0]], [[0, 0, -1]])
    near = int(np.flatnonzero(sc.tris["v0"][:, 2] == 0.5)[0])
    assert t[0] == 0.5 and tri[0] == near


def test_shadow_predicate(oracle):
    sc = one_triangle_scene()
    o, d = [[0.25, 0.25, 1.0]] * 5, [[0, 0, -1]] * 5
    dist = np.array([-1.0, 2.0, 1.0, 1.0 + 1e-5, 0.5], np.float32)            # closest hit at t = 1
    occ = oracle.occluded(sc, o, d, dist)
    # directional: any hit; point/emissive: t < dist - 2e-6 (pt.wgsl:394, :423, :465)
    assert occ.tolist() == [1, 1, 0, 1, 0]


def test_traversal_counters_cornell(oracle):
    """The reference traversal has no distance cull: a primary ray visits ~20+ nodes."""
    sc = scenes.make("cornell")
    cam = layout.make_camera(64, 64)
    ys, xs = np.mgrid[0:64, 0:64]
    o, d, _ = oracle.raygen(cam, xs.ravel(), ys.ravel(), np.zeros(4096, np.uint32))
    t, tri, _, _, st = oracle.intersect(sc, o, d)
    assert (t > 0).mean() > 0.9
    assert 15 < st.nodes_visited / 4096 < 80 and 3 < st.tris_tested / 4096 < 40
    assert st.max_stack <= 64


# ---- closed-form BSDF / sampling identities -------------------------------------------------
def test_power_heuristic(oracle):
    for f, g in ((0.3, 0.7), (5.0, 0.01), (1.0, 1.0)):
        a, b = oracle.power_heuristic(1, f, 1, g), oracle.power_heuristic(1, g, 1, f)
        assert abs(a + b - 1.0) < 1e-6 and abs(a - f * f / (f * f + g * g)) < 1e-6


def test_ggx_normalisation(oracle):
    """Integral of D(h) cos(theta_h) over the hemisphere = 1 (pt.wgsl:316-325, alpha = roughness^2)."""
    n = np.array([0, 0, 1], np.float32)
    for rough in (0.3, 0.5, 0.9):
        th = (np.arange(4000) + 0.5) / 4000 * (math.pi / 2)
        vals = [oracle.distribution_ggx(n, [math.sin(t), 0.0, math.cos(t)], rough) for t in th]
        integ = np.sum(np.array(vals) * np.cos(th) * np.sin(th)) * (math.pi / 2 / 4000) * 2 * math.pi
        assert abs(integ - 1.0) < 5e-3, (rough, integ)


def test_cosine_and_ggx_sampling(oracle):
    zs, st = [], 99
    for _ in range(20000):
        d, st = oracle.cosine_direction(st)
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and d[2] >= 0
        zs.append(d[2])
    assert abs(np.mean(zs) - 2 / 3) < 0.01                     # E[cos] under a cosine pdf
    n = np.array([0.0, 1.0, 0.0], np.float32)
    hs, st = [], 5
    for _ in range(5000):
        h, st = oracle.sample_ggx_normal(st, n, 0.5)
        assert abs(np.linalg.norm(h) - 1) < 1e-5
        hs.append(h @ n)
    assert min(hs) >= -1e-6                                     # half vectors stay in the normal's hemisphere


def test_eval_bsdf_branches(oracle):
    n, v = [0, 0, 1], [0, 0, 1]
    l = [math.sin(0.5), 0, math.cos(0.5)]
    # dielectric, no metal: pdf = cos/pi (diffuse lobe probability 1), f*cos >= diffuse term
    r = oracle.eval_bsdf([0.8, 0.8, 0.8], 0.5, 0.0, 0.0, 1.5, n, v, l)
    assert abs(r[3] - math.cos(0.5) / math.pi) < 1e-6
    assert (r[:3] > 0).all() and (r[:3] < 1).all()
    # transmission > 0: ((1 - F) * albedo, (1 - metallic) * transmission) whatever the directions (pt.wgsl:581-594)
    r = oracle.eval_bsdf([0.9, 1.0, 0.9], 0.1, 0.0, 1.0, 1.5, n, v, l)
    f0 = ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2
    assert abs(r[3] - 1.0) < 1e-6 and abs(r[1] - (1 - f0)) < 1e-6
    # light below the horizon: f*cos = 0, pdf floored at 1e-6 (pt.wgsl:613)
    r = oracle.eval_bsdf([0.8, 0.8, 0.8], 0.5, 0.0, 0.0, 1.5, n, v, [0, 0.6, -0.8])
    assert (r[:3] == 0).all() and r[3] == np.float32(1e-6)


# ---- closed-form renders ----------------------------------------------------------------------
def emissive_wall_scene():
    """A big emissive quad 2 units in front of the camera: every path ends at bounce 0 with
    radiance E * strength / (1 + t^2) (pt.wgsl:652-658)."""
    q = scenes._quad((-50, -50, -2), (50, -50, -2), (50, 50, -2), (-50, 50, -2), (0, 0, 1), 0)
    mats = np.array([scenes._material((0.8, 0.8, 0.8), emission=(1.0, 0.5, 0.25), strength=3.0)], layout.MATERIAL)
    return scenes._finish("wall", [q], mats)


def test_emission_furnace_closed_form(oracle):
    sc = emissive_wall_scene()
    cam = layout.make_camera(16, 16, position=(0, 0, 0), aperture=0.0)
    out, st = oracle.render(sc, cam, 1)
    assert st.segments == 256 and st.shadow_rays == 0
    ys, xs = np.mgrid[0:16, 0:16]
    o, d, _ = oracle.raygen(cam, xs.ravel(), ys.ravel(), np.zeros(256, np.uint32))
    t = (-2.0 / d[:, 2]).reshape(16, 16)
    want = np.minimum(np.array([1.0, 0.5, 0.25])[None, None] * 3.0 / (1 + t[..., None] ** 2), 2.5)
    assert np.allclose(out[..., :3], want, rtol=2e-6)
    assert (out[..., 3] == 0).all()


def test_running_mean_and_frame_order(oracle):
    sc = scenes.make("cornell")
    cam = layout.make_camera(24, 16)
    full, _ = oracle.render(sc, cam, 5)
    step = np.zeros_like(full)
    for f in range(5):                                           # the reference's per-frame loop (renderer.ts:415-454)
        step, _ = oracle.render(sc, layout.make_camera(24, 16, frame_index=f), 1, out=step)
    assert np.array_equal(full.view(np.uint32), step.view(np.uint32))
    # frame 0 overwrites whatever is in the buffer (pt.wgsl:754)
    junk = np.full_like(full, 7.0)
    f0, _ = oracle.render(sc, cam, 1, out=junk)
    f0b, _ = oracle.render(sc, cam, 1)
    assert np.array_equal(f0, f0b)
    # threads and row bands do not change a single bit
    t1, _ = oracle.render(sc, cam, 2, threads=1)
    t8, _ = oracle.render(sc, cam, 2, threads=8)
    assert np.array_equal(t1, t8)
    band = np.zeros_like(t1)
    for y0, y1 in ((0, 5), (5, 11), (11, 16)):
        band, _ = oracle.render(sc, cam, 2, out=band, y0=y0, y1=y1)
    assert np.array_equal(band, t1)
    assert (t1[..., :3].max() <= 2.5)                            # per-sample clamp (pt.wgsl:751)


def test_mis_on_off_same_mean(oracle):
    """MIS changes the estimator, not the expectation (here: not exactly — Appendix D-1: BSDF-sampled
    emission carries no MIS weight, so MIS-on counts direct light by both strategies). The check is
    therefore one-sided: both are positive, finite, and MIS-on >= MIS-off in mean."""
    sc = scenes.make("cornell")
    cam = layout.make_camera(48, 48)
    on, _ = oracle.render(sc, cam, 24, do_mis=1)
    off, _ = oracle.render(sc, cam, 24, do_mis=0)
    m_on, m_off = on[..., :3].mean(), off[..., :3].mean()
    assert np.isfinite(on).all() and np.isfinite(off).all()
    assert m_off > 0.02 and m_on > m_off


@pytest.mark.parametrize("name", ["cornell", "feature_box"])
def test_contract_vs_literal_build(oracle, oracle_strict, name):
    """The arithmetic contract (fused dot/cross, reciprocal multiply, polynomial sin/cos) against the
    literal transcription (IEEE '/', no FMA, libm): the same image up to discrete Monte-Carlo flips.

    Measured: the two builds agree on the RNG stream and to ~1e-7 on every ray, but
    the reference's light-visibility test `t < dist - 2e-6` (pt.wgsl:465) compares two numbers that
    differ by ~1e-6 (the shadow origin offset) at distances where one f32 ulp is 1-2.4e-7, so a few
    percent of next-event samples flip between "lit" and "self-occluded by the light" with the
    rounding of t and dist. Each flip is one small NEE term, so pixels differ by little and the image
    mean by ~0.3% (the fused position of the contract build yields slightly fewer false occlusions).
    Bar: means within 1%, mean |diff| below 1% of the mean, >= 90% of pixels within 5%."""
    sc = scenes.make(name)
    cam = layout.make_camera(64, 64)
    a, sa = oracle.render(sc, cam, 8)
    b, sb = oracle_strict.render(sc, cam, 8)
    d = np.abs(a - b)[..., :3]
    close = (d <= 1e-4 + 5e-2 * np.abs(b[..., :3])).all(axis=-1)
    assert close.mean() > 0.90, close.mean()
    assert abs(a[..., :3].mean() / b[..., :3].mean() - 1) < 0.01
    assert d.mean() < 0.01 * b[..., :3].mean()
    assert abs(sa.segments / sb.segments - 1) < 0.002
    # ray generation differs only in the last bits
    ys, xs = np.mgrid[0:64, 0:64]
    oa, da, ra = oracle.raygen(cam, xs.ravel(), ys.ravel(), np.zeros(4096, np.uint32))
    ob, db, rb = oracle_strict.raygen(cam, xs.ravel(), ys.ravel(), np.zeros(4096, np.uint32))
    assert np.array_equal(ra, rb) and np.abs(da - db).max() < 1e-6
    # closest hits: same triangles on >= 99.9% of rays, t within a few ulp
    ta, tria, _, _, _ = oracle.intersect(sc, oa, da)
    tb, trib, _, _, _ = oracle_strict.intersect(sc, oa, da)
    same = tria == trib
    assert same.mean() > 0.999
    assert np.allclose(ta[same], tb[same], rtol=1e-5)


def test_trace_path_log(oracle):
    sc = scenes.make("cornell")
    cam = layout.make_camera(32, 32)
    rad, log = oracle.trace_path(sc, cam, 16, 10, 0)
    assert 2 <= len(log) <= 9 and log[-1][15] == 0.0 and (log[:-1, 15] == 1.0).all()
    out, _ = oracle.render(sc, cam, 1)
    assert np.array_equal(np.minimum(rad, np.float32(2.5)), out[10, 16, :3])


@pytest.mark.parametrize("name,mis,bounces,ap", [("cornell", 1, 8, 0.001), ("cornell", 0, 4, 0.0), ("cornell_glass", 1, 8, 0.0),
                                                 ("feature_box", 1, 8, 0.05), ("cornell_spheres", 1, 8, 0.001),
                                                 ("random_soup", 1, 8, 0.02)])
def test_literal_transcription_equals_the_strict_build(oracle_strict, oracle_literal, scene_factory, name, mis, bounces, ap):
    """Two literal restatements of pt.wgsl that share no code — pt_oracle.c's PT_STRICT build (the contract build's control flow
    with literal arithmetic: the shading state rebuilt once for the winning triangle, RNG by pointer, guarded 1024-entry stack)
    and oracle/pt_literal.c (the reference's own shape: a whole HitInfo per accepted candidate, sampleLight tracing its own
    shadow ray, private RNG state, 64-entry stack) — must agree on EVERY bit: radiance, counters, camera rays, RNG states,
    closest hits and shadow predicates. A mis-reading of the WGSL would have to be made twice, in two different structures."""
    assert oracle_literal.literal and oracle_strict.strict and not oracle_strict.literal
    sc = scene_factory(name) if name != "random_soup" else __import__("ptmi.scenes", fromlist=["x"]).random_soup(7)
    W, H, frames = 56, 40, 3
    cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
    a, sa = oracle_strict.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
    b, sb = oracle_literal.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
    assert (sa.segments, sa.shadow_rays, sa.paths, sa.closest_hits, sa.nodes_visited, sa.tris_tested) == \
           (sb.segments, sb.shadow_rays, sb.paths, sb.closest_hits, sb.nodes_visited, sb.tris_tested)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a.view(np.uint32) != b.view(np.uint32)).sum()} radiance words differ"
    ys, xs = np.mgrid[0:H, 0:W]
    fr = (np.arange(W * H) % 5).astype(np.uint32)
    oa, da, ra = oracle_strict.raygen(cam, xs.ravel(), ys.ravel(), fr)
    ob, db, rb = oracle_literal.raygen(cam, xs.ravel(), ys.ravel(), fr)
    assert np.array_equal(ra, rb) and np.array_equal(oa.view(np.uint32), ob.view(np.uint32)) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
    ta, tria, ua, va, _ = oracle_strict.intersect(sc, oa, da)
    tb, trib, ub, vb, _ = oracle_literal.intersect(sc, oa, da)
    assert np.array_equal(tria, trib)
    for x, y in ((ta, tb), (ua, ub), (va, vb)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    dist = np.where(ta > 0, ta + np.float32(1e-6) * (np.arange(len(ta)) % 5), np.float32(-1.0)).astype(np.float32)
    assert np.array_equal(oracle_strict.occluded(sc, oa, da, dist), oracle_literal.occluded(sc, oa, da, dist))
