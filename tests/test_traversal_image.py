"""The traversal image ptmi_upload_scene derives from an uploaded BVH — host logic, no GPU: the hierarchy rebuilt over the
reference's leaves (csrc/fast_tree.hip) and its quantised form for the global traversal variant. The kernels' results
rest on two properties checked here on the image itself: every quantised child box, decoded with the kernel's own
fmaf, contains the exact box it stands for (so the descent cannot lose a leaf the reference's traversal,
src/shader/pt.wgsl:248-291, would test), and every leaf record carries the reference node's own box and its triangles
bit for bit (so the leaf is then tested exactly as pt.wgsl:266 tests it)."""
import numpy as np
import pytest

from ptmi import native, scenes


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "feature_box", "soup3", "soup8", "grid"])
def test_quantised_image_contains_the_exact_boxes(name):
    if name.startswith("soup"):
        sc = scenes.random_soup(int(name[4:]), n_tris=900)
    elif name == "grid":
        sc = scenes.grid_1m(n=96)                      # the 1 M-triangle scene's construction at 18 050 triangles
    else:
        sc = scenes.make(name)
    r = native.image_stats(sc)
    n_leaves = int((sc.nodes["triangle_count"] > 0).sum())
    assert r["leaves"] == n_leaves and r["wide_nodes"] == n_leaves - 1
    assert r["quantised_nodes"] == r["wide_nodes"]
    assert r["containment_violations"] == 0 and r["stream_mismatches"] == 0
    assert r["stream_dwords"] == 8 * n_leaves + 9 * len(sc.tris)
    assert 0 <= r["mean_area_growth"] < 0.05


def test_scene_with_a_huge_range_of_scales_keeps_the_exact_image():
    """A chain of boxes shrinking geometrically: one 16-bit grid over the scene cannot resolve the small ones; the image
    is then not quantised (the kernels walk the exact 64-byte nodes)."""
    r = native.image_stats(scenes.make("deep_chain"))
    assert r["wide_nodes"] > 0 and r["quantised_nodes"] == 0 and r["stream_dwords"] == 0


def test_flat_scene_quantises_with_a_zero_scale():
    """All triangles in one plane: that axis has extent 0, every plane number on it is 0."""
    t = scenes._quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1, 0), 0)
    parts = []
    for i in range(6):
        q = t.copy()
        for k in ("v0", "v1", "v2"):
            q[k][:, 0] += 3.0 * i
        parts.append(q)
    sc = scenes._finish("flat", parts, [scenes._material()])
    r = native.image_stats(sc)
    assert r["containment_violations"] == 0 and r["stream_mismatches"] == 0
    assert r["quantised_nodes"] == r["wide_nodes"] > 0
