"""The Node host's glTF path: host/gltf.js + host/scene_prep.js restate
src/renderer/loader.ts + src/renderer/gpu.ts:67-421 on top of the C++ BVH builder. Checked against a
float64 numpy model of the same transforms on a synthetic TRS hierarchy, on the reference's own .glb
assets when they are present (CPU only; never on the GPU box), and end to end against the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from ptmi import glb_io, layout, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "wgpu-path-tracing_amd", "host")
NODE = shutil.which("node")
REF_MODELS = "/root/reference/public/models"
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wgpu-path-tracing_amd"), "all"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(HOST, "addon")], stdout=subprocess.DEVNULL)


def quat(axis, angle):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    return [*(a * np.sin(angle / 2)), float(np.cos(angle / 2))]


def trs(t=(0, 0, 0), r=(0, 0, 0, 1), s=(1, 1, 1)):
    x, y, z, w = r
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4)
    M[:3, :3] = R * np.asarray(s, np.float64)[None, :]
    M[:3, 3] = t
    return M


def synthetic_glb(path):
    """A small room: floor + emissive ceiling panel + a cube parented under a rotated, scaled node
    (child -> parent -> grandparent chain), a metal sphere, a point and a directional light."""
    box = scenes._box((0, 0, 0), (1, 1, 1), 0)
    sph = scenes._uv_sphere((0, 0, 0), 0.5, 0, segments=10, rings=6)

    def mesh_of(tris, material):
        pos = np.concatenate([tris["v0"], tris["v1"], tris["v2"]]).reshape(3, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
        nrm = np.concatenate([tris["n0"], tris["n1"], tris["n2"]]).reshape(3, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
        uv = np.concatenate([tris["uv0"], tris["uv1"], tris["uv2"]]).reshape(3, -1, 2).transpose(1, 0, 2).reshape(-1, 2)
        return {"positions": pos, "normals": nrm, "uvs": uv, "indices": np.arange(len(pos)), "material": material}

    floor = scenes._quad((-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2), (0, 1, 0), 0)
    panel = scenes._quad((-0.5, 2.5, -0.5), (0.5, 2.5, -0.5), (0.5, 2.5, 0.5), (-0.5, 2.5, 0.5), (0, -1, 0), 0)
    meshes = [mesh_of(floor, 0), mesh_of(panel, 1), mesh_of(box, 2), mesh_of(sph, 3), mesh_of(box, None)]
    materials = [
        {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1], "metallicFactor": 0, "roughnessFactor": 0.5}},
        {"emissiveFactor": [1, 0.9, 0.7], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.0}},
         "pbrMetallicRoughness": {"metallicFactor": 0, "roughnessFactor": 0.5}},
        {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.2, 0.2, 1], "metallicFactor": 0, "roughnessFactor": 0.4},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.0}, "KHR_materials_ior": {"ior": 1.45}}},
        {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.5, 1], "roughnessFactor": 0.2}},   # metallic defaults to 1
    ]
    g, p = dict(translation=[0.3, 0.0, -0.4], rotation=quat((0, 1, 0), 0.6), scale=[1.2, 1.0, 0.8]), \
        dict(translation=[0.0, 0.4, 0.2], rotation=quat((1, 0, 1), 0.3), scale=[0.7, 0.7, 0.7])
    c = dict(translation=[0.1, 0.5, 0.0], rotation=quat((0, 0, 1), -0.4), scale=[0.6, 1.1, 0.6])
    nodes = [
        {"mesh": 0}, {"mesh": 1},
        {**g, "children": [3]}, {**p, "children": [4]}, {**c, "mesh": 2},                      # grandparent -> parent -> cube
        {"mesh": 3, "translation": [-0.9, 0.5, 0.5]},
        {"mesh": 4, "translation": [1.2, 0.25, 0.9], "scale": [0.5, 0.5, 0.5]},                 # primitive without a material
        {"light": 0, "translation": [0.0, 1.8, 1.5]},
        {"light": 1, "rotation": quat((1, 0, 0), -0.9)},
    ]
    lights = [{"type": "point", "color": [1, 0.8, 0.6], "intensity": 3.0}, {"type": "directional", "intensity": 0.5}]
    glb_io.write_glb(path, meshes, nodes, materials, lights)
    world = {0: np.eye(4), 1: np.eye(4), 4: trs(g["translation"], g["rotation"], g["scale"]) @ trs(p["translation"], p["rotation"], p["scale"])
             @ trs(c["translation"], c["rotation"], c["scale"]), 5: trs([-0.9, 0.5, 0.5]), 6: trs([1.2, 0.25, 0.9], s=[0.5, 0.5, 0.5])}
    return meshes, nodes, world


def prepare(glb, out):
    os.makedirs(out, exist_ok=True)
    subprocess.check_call([NODE, os.path.join(HOST, "prepare_cli.js"), str(glb), str(out)], stdout=subprocess.DEVNULL)
    return glb_io.load_blob_dir(str(out)), json.load(open(os.path.join(out, "info.json")))


def test_synthetic_hierarchy_matches_float64_model(tmp_path):
    _build()
    meshes, nodes, world = synthetic_glb(tmp_path / "s.glb")
    sc, info = prepare(tmp_path / "s.glb", tmp_path / "out")
    n_tri = sum(len(m["indices"]) // 3 for m in meshes)
    assert info["counts"] == {"triangles": n_tri, "materials": 5, "bvhNodes": len(sc.nodes), "lights": len(sc.lights),
                              "punctualLights": 2}
    # every expected world-space triangle (float64 model) is present among the prepared ones, per material
    for node_i, W in world.items():
        m = meshes[nodes[node_i]["mesh"]]
        mat_index = [0, 1, 4, 5, 6].index(node_i)          # one material per primitive, in node order (gpu.ts:285-291)
        pos = (np.c_[m["positions"], np.ones(len(m["positions"]))] @ W.T)[:, :3].reshape(-1, 3, 3)
        nm = np.linalg.inv(W[:3, :3]).T
        nrm = m["normals"] @ nm.T
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = nrm.reshape(-1, 3, 3)
        got = sc.tris[sc.tris["material_index"] == mat_index]
        assert len(got) == len(pos)
        gp = np.stack([got["v0"], got["v1"], got["v2"]], 1).astype(np.float64)
        gn = np.stack([got["n0"], got["n1"], got["n2"]], 1).astype(np.float64)
        key = lambda a: np.lexsort(np.round(a.reshape(len(a), -1), 4).T[::-1])
        ia, ib = key(gp), key(pos)
        assert np.abs(gp[ia] - pos[ib]).max() < 2e-6
        assert np.abs(gn[ia] - nrm[ib]).max() < 2e-6
    # materials: glTF defaults and extensions (gpu.ts:356-421)
    mt = sc.mats
    assert np.allclose(mt["base_color"][0], 0.8) and mt["metallic"][3] == 1.0 and np.isclose(mt["roughness"][3], 0.2)
    assert np.isclose(mt["emissive_strength"][1], 12.0) and np.isclose(mt["ior"][2], 1.45) and mt["ior"][0] == 1.5
    assert np.isclose(mt["roughness"][4], 0.1) and mt["metallic"][4] == 0.0 and (mt["base_color"][4] == 1.0).all()   # no material
    # lights: punctual first in node order, then emissive triangles in post-sort order (gpu.ts:105-138)
    lt = sc.lights
    assert lt["light_type"][:2].tolist() == [layout.LIGHT_POINT, layout.LIGHT_DIRECTIONAL]
    assert np.allclose(lt["position"][0], [0.0, 1.8, 1.5]) and np.isclose(lt["intensity"][0], 3.0)
    assert np.allclose(lt["position"][1], [0, -np.sin(0.9), -np.cos(0.9)], atol=1e-6)       # (0,0,-1) rotated by the node
    em = np.flatnonzero(sc.tris["material_index"] == 1)
    assert lt["triangle_index"][2:].tolist() == em.tolist() and (lt["light_type"][2:] == 0).all()
    # the BVH the host built is the reference builder's tree over the sorted triangles
    from test_scene_host import _check_tree
    assert _check_tree(sc.nodes, sc.tris) == info["bvhDepth"]


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference assets are not on this machine")
@pytest.mark.parametrize("name,tris,mats", [("cornell2", 1004, 7), ("transform", 40, 5), ("monkey", 970, 2),
                                            ("metal", 5074, 8), ("glass_box", 11790, 15), ("untitled", 2882, 4)])
def test_reference_assets_load(tmp_path, name, tris, mats):
    """The reference's own sample scenes go through the host unchanged (read in place, never copied)."""
    _build()
    sc, info = prepare(os.path.join(REF_MODELS, name + ".glb"), tmp_path / name)
    assert (len(sc.tris), len(sc.mats)) == (tris, mats)
    from test_scene_host import _check_tree
    _check_tree(sc.nodes, sc.tris)
    assert len(sc.lights) >= 2 and np.isfinite(sc.tris["v0"]).all()
    n = np.stack([sc.tris["n0"], sc.tris["n1"], sc.tris["n2"]], 1)
    assert np.abs(np.linalg.norm(n, axis=-1) - 1).max() < 1e-4


@pytest.mark.gpu
def test_glb_render_through_node_matches_oracle(tmp_path, oracle):
    """.glb -> JS host (parse, prepareScene, BVH) -> N-API -> HIP, against the oracle on the blobs the host built."""
    _build()
    synthetic_glb(tmp_path / "s.glb")
    sc, _ = prepare(tmp_path / "s.glb", tmp_path / "out")
    W, H = 96, 64
    out = subprocess.check_output([NODE, os.path.join(HOST, "render_cli.js"), str(tmp_path / "s.glb"), str(tmp_path / "o.f32"),
                                   "--width", str(W), "--height", str(H), "--frames", "6", "--batch", "3"], text=True)
    st = json.loads(out.strip().splitlines()[-1])
    got = np.fromfile(tmp_path / "o.f32", np.float32).reshape(H, W, 4)
    ref, ost = oracle.render(sc, layout.make_camera(W, H), 6)
    assert st["segments"] == ost.segments and st["shadowRays"] == ost.shadow_rays
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert got[..., :3].mean() > 0.005
