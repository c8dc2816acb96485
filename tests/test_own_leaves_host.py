"""The library's own leaves (ptmi_options.leaves = 2), host side — no GPU.

(1) The image ptmi_upload_scene builds (csrc/fast_tree.hip pt_build_own_tree, through the host-only ptmi_debug_build_image): every
triangle some reference leaf lists appears exactly once, in a leaf of at most leaf_tris triangles; every box contains the triangles
below it with the promised padding; the quantised planes contain the padded boxes when decoded with the kernels' own fmaf; the
per-triangle table holds the box of the reference leaf that lists the triangle (src/renderer/bvh.ts:86-127 builds those leaves).
(2) The traversal over that image (tools/own_sim.c: the per-ray arithmetic of csrc/traverse_own.hip, replayed on the CPU) returns the
reference traversal's (t, triangle) and shadow verdicts — src/shader/pt.wgsl:248-291, :394/:423/:465 — on every ray of real renders
(recorded by the oracle: camera, bounce and shadow rays) and on irregular / far-away rays, with exact and with quantised nodes.
The same comparison on 10^8 rays: tools/own_leaf_gate.py (profiles/r04_own_leaves/)."""
import os
import sys

import numpy as np
import pytest

from ptmi import layout, native, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import own_leaf_gate as gate            # noqa: E402

REF_LEAF = 0x80000000


def scene_of(name):
    if name.startswith("soup"):
        return scenes.random_soup(int(name[4:]), n_tris=900)
    if name == "grid":
        return scenes.grid_1m(n=96)                       # the 1 M-triangle scene's construction at 18 050 triangles
    return scenes.make(name)


def children(wn):
    """[(node, side, lo[3], hi[3], ref)] of an [n, 16] wide-node array"""
    u = wn.view(np.uint32)
    for i in range(len(wn)):
        w = wn[i]
        yield i, 0, w[0:3], w[3:6], int(u[i, 12])
        yield i, 1, w[6:9], w[9:12], int(u[i, 13])


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "feature_box", "soup3", "grid"])
@pytest.mark.parametrize("k", [1, 2, 4])
def test_own_image_lists_every_triangle_once_inside_padded_boxes(name, k):
    sc = scene_of(name)
    info, wn, qn, tp, lb = native.build_image(sc, leaves=2, leaf_tris=k)
    assert info.leaves_used == 2 and info.max_leaf_tris <= k and info.pad > 0
    # the triangles of the reachable reference leaves, each once, with the contract's image (v0, v1 - v0, v2 - v0)
    listed = np.zeros(len(sc.tris), bool)
    for n in sc.nodes[sc.nodes["triangle_count"] > 0]:
        listed[n["triangle_offset"]:n["triangle_offset"] + n["triangle_count"]] = True
    orig = tp[:, 3].copy().view(np.uint32)
    assert len(orig) == listed.sum() and np.array_equal(np.sort(orig), np.flatnonzero(listed))
    T = sc.tris[orig]
    assert np.array_equal(tp[:, 0:3], T["v0"]) and np.array_equal(tp[:, 4:7], T["v1"] - T["v0"]) and np.array_equal(tp[:, 8:11], T["v2"] - T["v0"])
    # the per-triangle table: the box of the reference leaf that lists it
    for n in sc.nodes[sc.nodes["triangle_count"] > 0][:200]:
        s = slice(n["triangle_offset"], n["triangle_offset"] + n["triangle_count"])
        assert (lb[s, 0:3] == n["aabb_min"]).all() and (lb[s, 4:7] == n["aabb_max"]).all()
    # leaves partition the image; every box holds its triangles' vertices with the padding to spare; depth as reported
    seen = np.zeros(len(tp), np.int32)
    verts = np.stack([T["v0"], T["v1"], T["v2"]], axis=1).astype(np.float64)            # [m, 3, 3]

    def walk(ref, depth):
        """(lo, hi, depth) of the exact bounds below a child reference"""
        if ref & REF_LEAF:
            first, cnt = ref & ((1 << 26) - 1), ((ref >> 26) & 31) + 1
            assert cnt <= k
            seen[first:first + cnt] += 1
            v = verts[first:first + cnt].reshape(-1, 3)
            return v.min(axis=0), v.max(axis=0), depth
        w = wn[ref]
        u = wn.view(np.uint32)[ref]
        l0, h0, d0 = walk(int(u[12]), depth + 1)
        l1, h1, d1 = walk(int(u[13]), depth + 1)
        for lo, hi, blo, bhi in ((l0, h0, w[0:3], w[3:6]), (l1, h1, w[6:9], w[9:12])):
            assert (blo.astype(np.float64) <= lo - 0.99 * info.pad).all() and (bhi.astype(np.float64) >= hi + 0.99 * info.pad).all()
        return np.minimum(l0, l1), np.maximum(h0, h1), max(d0, d1)

    sys.setrecursionlimit(10000)
    lo, hi, depth = walk(info.root_ref, 1)
    assert (seen == 1).all()
    assert depth == info.depth and depth <= (14 if len(tp) <= 2048 else 60)
    assert (np.array(info.root_min[:]) <= lo - 0.99 * info.pad).all() and (np.array(info.root_max[:]) >= hi + 0.99 * info.pad).all()
    # the quantised planes, decoded with the kernel's fmaf (float32 fused multiply-add == exact product rounded once: done in float64
    # here, whose 53 bits hold the 24 x 16-bit product and the sum exactly enough to round to the same float), contain the padded boxes
    assert info.quantised and qn is not None
    # the two images number their nodes differently: walk them together
    todo = [(0, 0)] if not (info.root_ref & REF_LEAF) else []
    qo, qs = np.array(info.q_origin[:], np.float64), np.array(info.q_scale[:], np.float64)
    n_seen = 0
    while todo:
        i, j = todo.pop()
        n_seen += 1
        u = wn.view(np.uint32)[i]
        for side in range(2):
            q = qn[j, 4 * side:4 * side + 4]
            planes = np.array([q[0] & 0xFFFF, q[0] >> 16, q[1] & 0xFFFF, q[1] >> 16, q[2] & 0xFFFF, q[2] >> 16], np.float64)
            dlo = (qs * planes[0:3] + qo).astype(np.float32)
            dhi = (qs * planes[3:6] + qo).astype(np.float32)
            blo, bhi = wn[i, 6 * side:6 * side + 3], wn[i, 6 * side + 3:6 * side + 6]
            assert (dlo <= blo).all() and (dhi >= bhi).all()
            ref = int(u[12 + side])
            if ref & REF_LEAF:
                assert int(q[3]) == ref
            else:
                todo.append((ref, int(q[3])))
    assert n_seen == len(wn)


def special_rays(sc, n, seed):
    """Irregular directions (zeros, subnormals, tiny components), origins on box planes and vertices, far-away origins: the rays the
    own image hands to the uploaded tree."""
    rng = np.random.default_rng(seed)
    corners = np.concatenate([sc.nodes["aabb_min"], sc.nodes["aabb_max"], sc.tris["v0"], sc.tris["v1"]])
    o = corners[rng.integers(0, len(corners), n)].astype(np.float32)
    o[::3] += (rng.standard_normal((len(o[::3]), 3)) * 0.3).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    kind, axis, rows = rng.integers(0, 7, n), rng.integers(0, 3, n), np.arange(n)
    d[kind == 0] = 0.0
    d[rows[kind == 0], axis[kind == 0]] = rng.choice([-1.0, 1.0], int((kind == 0).sum()))
    d[rows[kind == 1], axis[kind == 1]] = 0.0
    d[rows[kind == 2], axis[kind == 2]] = -0.0
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30).astype(np.float32)
    d[rows[kind == 3], axis[kind == 3]] = np.float32(1e-41)              # subnormal: 1/d = inf
    d[rows[kind == 4], axis[kind == 4]] = np.float32(1e-20)              # 1/d beyond 2^60: outside what the padding is proven for
    far = kind == 5                                                      # origins far outside the scene, aimed back at it
    o[far] = (o[far] + d[far] * np.float32(-300.0)).astype(np.float32)
    rec = np.zeros((n, 9), np.float32)
    rec[:, 0:3], rec[:, 3:6] = o, d
    dist = (rng.random(n) * 2.0).astype(np.float32)
    dist[::3] = -1.0
    dist[1::3] = 0.0                                                     # a third closest-hit rays
    rec[:, 6] = dist
    return rec


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "feature_box", "soup3", "soup8", "grid"])
def test_own_traversal_returns_the_reference_traversal(name, oracle):
    sc = scene_of(name)
    L = gate.sim_lib()
    W, H = 240, 136
    cam = layout.make_camera(W, H, aperture=0.01, focus_distance=2.8)
    rec, n = gate.tap_rays(oracle, sc, cam, 2, 0, H, 1 << 21)
    assert n == len(rec) and (rec[:, 6] != 0).sum() > 1000 and (rec[:, 7] > 0).mean() > 0.3
    # the special rays get their reference results from the oracle's own entry points
    sp = special_rays(sc, 40_000, 17)
    t, tri, _, _, _ = oracle.intersect(sc, sp[:, 0:3], sp[:, 3:6])
    sp[:, 7] = t
    sp[:, 8] = tri.view(np.float32)
    rec = np.ascontiguousarray(np.concatenate([rec, sp]))
    for k in (2, 4):
        img = gate.Image(sc, 2, k)
        for quant in ((0, 1) if img.qn is not None else (0,)):
            for cull, deferred in ((1, 0), (1, 1), (0, 0)):
                sums, diff = gate.run(L, img, rec, quant, cull, deferred, want_diff=8)
                assert int(sums[8]) == 0 and int(sums[9]) == 0, (name, k, quant, cull, deferred, sums.tolist(), rec[diff.astype(np.int64)])
                assert int(sums[10]) > 5_000                              # the special rays went the slow way ...
                assert int(sums[11]) < 1e-4 * len(rec)                    # ... and almost no winner had to be traced again
    # the gate itself: a quarter of the triangle tests, fewer instructions by the estimate (65 / 54 per box pair, 54 per triangle)
    base, _ = gate.run(L, gate.Image(sc, 1), rec, 0)
    own, _ = gate.run(L, gate.Image(sc, 2), rec, 0)
    tris_ref, tris_own = int(base[2]) / int(base[6]), int(own[2]) / int(own[6])
    assert int(base[8]) == 0 and int(base[9]) == 0
    if name in ("cornell", "cornell_spheres", "grid"):
        assert tris_own < 0.5 * tris_ref, (tris_ref, tris_own)
        est = lambda s, box: box * int(s[0]) + 54 * int(s[2]) + 20 * int(s[1])
        assert est(own, 54) < 0.8 * est(base, 65)


def test_scene_with_a_non_finite_vertex_keeps_the_reference_leaves():
    sc = scenes.make("cornell")
    tris = sc.tris.copy()
    tris["v1"][5, 1] = np.inf
    bad = scenes.Scene("bad", tris, sc.mats, sc.nodes, sc.lights, sc.atlas, sc.bvh_depth)
    info, wn, qn, tp, lb = native.build_image(bad, leaves=2)
    assert info.leaves_used == 1 and lb is None
    info, *_ = native.build_image(sc, leaves=2, keep_reference_tree=1)
    assert info.leaves_used == 1
