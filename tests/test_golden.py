"""Committed fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py): complete inputs
and expected outputs. CPU: the oracle must reproduce them bit for bit on this machine (pins the
oracle against compiler / libm / CPU drift). GPU: the HIP path must reproduce them through the C ABI."""
import glob
import os

import numpy as np
import pytest

from ptmi import layout, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    z = np.load(path)                                   # allow_pickle stays False
    atlas = z["atlas"] if z["atlas"].size else None
    sc = scenes.Scene(os.path.basename(path), z["tris"].view(layout.TRIANGLE).copy(), z["mats"].view(layout.MATERIAL).copy(),
                      z["nodes"].view(layout.BVH_NODE).copy(), z["lights"].view(layout.LIGHT).copy(), atlas)
    cam = np.frombuffer(z["camera"].tobytes(), layout.CAMERA).copy().reshape(())
    return z, sc, cam


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_fixtures_present():
    assert len(FILES) >= 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    z, sc, cam = load(path)
    out, st = oracle.render(sc, cam, int(z["frames"]), max_bounces=int(z["bounces"]), do_mis=int(z["mis"]))
    assert st.segments == int(z["segments"]) and st.shadow_rays == int(z["shadow_rays"])
    assert same(out, z["image"])
    t, tri, u, v, _ = oracle.intersect(sc, z["ray_o"], z["ray_d"])
    assert np.array_equal(tri, z["hit_tri"]) and same(t, z["hit_t"]) and same(u, z["hit_u"]) and same(v, z["hit_v"])
    assert np.array_equal(oracle.occluded(sc, z["ray_o"], z["ray_d"], z["shadow_dist"]), z["shadow_occluded"])


def test_generators_reproduce_fixture_scenes():
    """The procedural scene + BVH build give the very blobs stored in the fixtures."""
    for path, name in ((FILES[0], "cornell"),):
        z, sc, _ = load(path)
        fresh = scenes.make(name)
        assert fresh.tris.tobytes() == sc.tris.tobytes() and fresh.nodes.tobytes() == sc.nodes.tobytes()
        assert fresh.lights.tobytes() == sc.lights.tobytes() and fresh.mats.tobytes() == sc.mats.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_reproduces_golden(gpu_ctx, path):
    z, sc, cam = load(path)
    gpu_ctx.upload_scene(sc)
    gpu_ctx.resize(int(cam["width"]), int(cam["height"]))
    gpu_ctx.set_options(max_bounces=int(z["bounces"]), do_mis=int(z["mis"]), tile_y0=0, tile_y1=0,
                        frames_per_batch=0, cull=1, traversal=0)
    gpu_ctx.reset_stats()
    gpu_ctx.dispatch(cam, int(z["frames"]))
    out = gpu_ctx.read_output()
    st = gpu_ctx.stats()
    assert st.segments == int(z["segments"]) and st.shadow_rays == int(z["shadow_rays"])
    assert same(out, z["image"])
    t, tri, u, v = gpu_ctx.debug_intersect(z["ray_o"], z["ray_d"])
    assert np.array_equal(tri, z["hit_tri"]) and same(t, z["hit_t"]) and same(u, z["hit_u"]) and same(v, z["hit_v"])
    assert np.array_equal(gpu_ctx.debug_occluded(z["ray_o"], z["ray_d"], z["shadow_dist"]), z["shadow_occluded"])
