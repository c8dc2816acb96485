"""Host-side scene preparation (libptmi_scene.so) against the reference's own unit-test
vectors (src/spec/arr.test.ts:4-44, the only tests the reference has) and against the
pure-Python restatement oracle/bvh_ref.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bvh_ref  # noqa: E402
from ptmi import layout, scene_host, scenes  # noqa: E402


# ---- the reference's golden vectors: src/spec/arr.test.ts -----------------------------
def test_arr_partial_range():                                   # arr.test.ts:5-9
    out = scene_host.sort_partially([5, 2, 8, 1, 9, 3, 7], 1, 4)
    assert out.tolist() == [5, 1, 2, 8, 9, 3, 7]
    assert bvh_ref.sort_partially([5, 2, 8, 1, 9, 3, 7], 1, 4, lambda a, b: a - b) == [5, 1, 2, 8, 9, 3, 7]


def test_arr_duplicates():                                      # arr.test.ts:11-15
    out = scene_host.sort_partially([3, 3, 2, 2, 1, 1], 0, 4)
    assert out.tolist() == [2, 2, 3, 3, 1, 1]
    assert bvh_ref.sort_partially([3, 3, 2, 2, 1, 1], 0, 4, lambda a, b: a - b) == [2, 2, 3, 3, 1, 1]


def test_arr_single_element_range():                            # arr.test.ts:17-21
    out = scene_host.sort_partially([5, 2, 8, 1, 9], 2, 3)
    assert out.tolist() == [5, 2, 8, 1, 9]


def test_arr_custom_compare():                                  # arr.test.ts:23-27 (reverse alphabetical)
    words = ["banana", "apple", "cherry", "date"]
    got = bvh_ref.sort_partially(list(words), 0, 3, lambda a, b: (b > a) - (b < a))
    assert got == ["cherry", "banana", "apple", "date"]
    rank = {w: i for i, w in enumerate(sorted(words))}          # same order through the f64 entry point
    out = scene_host.sort_partially([rank[w] for w in words], 0, 3, descending=True)
    inv = {i: w for w, i in rank.items()}
    assert [inv[int(i)] for i in out] == ["cherry", "banana", "apple", "date"]


def test_arr_invalid_indices():                                 # arr.test.ts:29-43
    for s, e in ((-1, 3), (3, 2), (0, 6)):
        with pytest.raises(scene_host.SceneError, match="Invalid indices"):
            scene_host.sort_partially([1, 2, 3, 4, 5], s, e)
        with pytest.raises(ValueError, match="Invalid indices"):
            bvh_ref.sort_partially([1, 2, 3, 4, 5], s, e, lambda a, b: a - b)


# ---- C++ restatement == Python restatement ----------------------------------------------
@pytest.mark.parametrize("n,seed", [(11, 0), (12, 1), (50, 2), (257, 3), (1000, 4)])
def test_sort_matches_python(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, max(2, n // 3), n).astype(np.float64)    # many duplicates: order of equals is the point
    s, e = (0, n) if seed % 2 == 0 else (n // 5, n - n // 7)
    got = scene_host.sort_partially(a.copy(), s, e)
    want = bvh_ref.sort_partially(list(a), s, e, lambda x, y: x - y)
    assert got.tolist() == want
    assert got[s:e].tolist() == sorted(a[s:e].tolist())


def _soup(n, seed, grid=False):
    rng = np.random.default_rng(seed)
    t = np.zeros(n, layout.TRIANGLE)
    c = rng.random((n, 3)).astype(np.float32) * 4 - 2
    if grid:                                                     # many equal centroids along each axis
        c = np.round(c * 2) / 2
    for k in ("v0", "v1", "v2"):
        t[k] = c + (rng.random((n, 3)).astype(np.float32) - 0.5) * (0.0 if grid and k == "v0" else 0.3)
    t["material_index"] = np.arange(n) % 3
    t["uv0"][:, 0] = np.arange(n)                               # identity tag to follow the permutation
    return t


@pytest.mark.parametrize("n,seed,grid", [(1, 0, False), (4, 1, False), (5, 2, False), (37, 3, False),
                                         (120, 4, True), (300, 5, False), (301, 6, True)])
def test_bvh_matches_python(n, seed, grid):
    tris = _soup(n, seed, grid)
    py_tris = [dict(v0=t["v0"].copy(), v1=t["v1"].copy(), v2=t["v2"].copy(), tag=int(t["uv0"][0])) for t in tris]
    py_nodes = bvh_ref.build_bvh(py_tris)
    nodes, depth = scene_host.build_bvh(tris)
    assert [int(t["uv0"][0]) for t in tris] == [t["tag"] for t in py_tris], "triangle order differs"
    assert len(nodes) == len(py_nodes)
    for a, b in zip(nodes, py_nodes):
        assert (int(a["left"]), int(a["right"]), int(a["triangle_offset"]), int(a["triangle_count"])) == \
               (b["left"], b["right"], b["offset"], b["count"])
        assert np.array_equal(a["aabb_min"], b["min"]) and np.array_equal(a["aabb_max"], b["max"])
    assert depth >= 1


def _check_tree(nodes, tris):
    seen = np.zeros(len(tris), int)
    stack = [(0, 1)]
    maxd = 0
    while stack:
        i, d = stack.pop()
        n = nodes[i]
        maxd = max(maxd, d)
        if n["triangle_count"] > 0:
            assert n["left"] == 0xFFFFFFFF and n["right"] == 0xFFFFFFFF          # bvh.ts:87-88
            assert n["triangle_count"] <= 4
            sl = slice(int(n["triangle_offset"]), int(n["triangle_offset"] + n["triangle_count"]))
            seen[sl] += 1
            pts = np.concatenate([tris[sl]["v0"], tris[sl]["v1"], tris[sl]["v2"]])
            assert np.array_equal(pts.min(0), n["aabb_min"]) and np.array_equal(pts.max(0), n["aabb_max"])
        else:
            l, r = int(n["left"]), int(n["right"])
            assert r == l + 1                                                       # bvh.ts:130-134
            for c in (l, r):
                assert (nodes[c]["aabb_min"] >= n["aabb_min"]).all() and (nodes[c]["aabb_max"] <= n["aabb_max"]).all()
                stack.append((c, d + 1))
    assert (seen == 1).all()
    return maxd


def test_cornell_scene_contract():
    sc = scenes.make("cornell")
    assert len(sc.tris) == 996 and len(sc.mats) == 7 and len(sc.lights) == 2
    assert _check_tree(sc.nodes, sc.tris) == sc.bvh_depth
    # gpu.ts:121-138: emissive lights in ascending post-sort triangle index, colour = emission
    li = sc.lights["triangle_index"]
    assert (np.diff(li.astype(int)) > 0).all()
    emissive = np.flatnonzero(np.linalg.norm(sc.mats["emission"][sc.tris["material_index"]], axis=1) > 0)
    assert li.tolist() == emissive.tolist()
    assert np.allclose(sc.lights["intensity"], 13.8) and (sc.lights["light_type"] == layout.LIGHT_EMISSIVE).all()
    # inward-facing walls: geometric normal of the floor points up (NEE needs front faces, pt.wgsl:661)
    t = sc.tris
    gn = np.cross(t["v1"] - t["v0"], t["v2"] - t["v0"])
    floor = ((np.abs(t["v0"][:, 1]) < 1e-6) & (np.abs(t["v1"][:, 1]) < 1e-6) & (np.abs(t["v2"][:, 1]) < 1e-6)
             & (np.abs(t["v0"][:, 0]) == 1.0))
    assert floor.sum() == 2 and (gn[floor][:, 1] > 0).all()


def test_other_scenes_build():
    fb = scenes.make("feature_box")
    assert set(fb.lights["light_type"].tolist()) == {0, 1, 2}
    assert fb.lights["light_type"][:2].tolist() == [layout.LIGHT_POINT, layout.LIGHT_DIRECTIONAL]   # punctual first
    assert fb.atlas.dtype == np.float16 and fb.atlas.shape == (256, 256, 4)
    _check_tree(fb.nodes, fb.tris)
    sp = scenes.make("cornell_spheres")
    assert len(sp.tris) == 996 + 3 * 960 and sp.atlas.shape == (1024, 1024, 4)
    _check_tree(sp.nodes, sp.tris)


def test_builder_rejects_non_finite():
    t = _soup(10, 0)
    t["v1"][3, 1] = np.nan
    with pytest.raises(scene_host.SceneError):
        scene_host.build_bvh(t)


@pytest.mark.parametrize("n,seed,grid", [(70_000, 11, False), (90_001, 12, True)])
def test_threaded_builder_is_byte_identical(n, seed, grid):
    """Ranges above 32 768 triangles are built by several threads and spliced in the reference's node order
    (left, right, everything under right, everything under left): same bytes as the reference's single loop."""
    base = _soup(n, seed, grid)
    one = base.copy()
    nodes1, depth1 = scene_host.build_bvh(one, threads=1)
    for threads in (0, 2, 5):
        many = base.copy()
        nodes, depth = scene_host.build_bvh(many, threads=threads)
        assert many.tobytes() == one.tobytes() and nodes.tobytes() == nodes1.tobytes() and depth == depth1
    assert _check_tree(nodes1, one) == depth1
