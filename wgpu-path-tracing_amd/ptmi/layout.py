"""numpy views of the boundary byte layouts of include/ptmi_layout.h.

These are the WGSL storage/uniform layouts of the reference's bind group 0
(reference: src/shader/pt.wgsl:7-78; packed on the host at
src/renderer/renderer.ts:242-355).
"""
import numpy as np

F, U = "<f4", "<u4"

RECT = np.dtype([("x", U), ("y", U), ("w", U), ("h", U)])

MATERIAL = np.dtype([
    ("base_color", F, 3), ("metallic", F), ("roughness", F), ("_pad0", F, 3),
    ("emission", F, 3), ("emissive_strength", F), ("ior", F), ("transmission", F),
    ("albedo_map", RECT), ("normal_map", RECT), ("pbr_map", RECT), ("emissive_map", RECT),
    ("_pad1", F, 2),
])

TRIANGLE = np.dtype([
    ("v0", F, 3), ("_p0", F), ("v1", F, 3), ("_p1", F), ("v2", F, 3), ("_p2", F),
    ("n0", F, 3), ("_p3", F), ("n1", F, 3), ("_p4", F), ("n2", F, 3), ("_p5", F),
    ("uv0", F, 2), ("uv1", F, 2), ("uv2", F, 2), ("material_index", U), ("_p6", U),
])

BVH_NODE = np.dtype([
    ("aabb_min", F, 3), ("_p0", F), ("aabb_max", F, 3), ("_p1", F),
    ("left", U), ("right", U), ("triangle_offset", U), ("triangle_count", U),
])

LIGHT = np.dtype([
    ("position", F, 3), ("light_type", U), ("color", F, 3), ("intensity", F),
    ("triangle_index", U), ("_pad", U, 3),
])

CAMERA = np.dtype([
    ("position", F, 3), ("_p0", F), ("forward", F, 3), ("_p1", F), ("right", F, 3), ("_p2", F),
    ("up", F, 3), ("fov", F), ("aspect", F), ("width", U), ("height", U), ("frame_index", U),
    ("aperture", F), ("focus_distance", F), ("_p3", U, 2),
])

LIGHT_EMISSIVE, LIGHT_DIRECTIONAL, LIGHT_POINT = 0, 1, 2

assert MATERIAL.itemsize == 128 and TRIANGLE.itemsize == 128
assert BVH_NODE.itemsize == 48 and LIGHT.itemsize == 48 and CAMERA.itemsize == 96
assert MATERIAL.fields["albedo_map"][1] == 56 and MATERIAL.fields["emissive_map"][1] == 104
assert TRIANGLE.fields["uv0"][1] == 96 and TRIANGLE.fields["material_index"][1] == 120
assert BVH_NODE.fields["left"][1] == 32 and BVH_NODE.fields["triangle_count"][1] == 44
assert LIGHT.fields["light_type"][1] == 12 and LIGHT.fields["triangle_index"][1] == 32
assert CAMERA.fields["fov"][1] == 60 and CAMERA.fields["focus_distance"][1] == 84


def make_camera(width, height, *, position=(0.0, 1.0, 2.8), forward=(0.0, 0.0, -1.0),
                right=(1.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=np.pi / 3, aspect=None,
                frame_index=0, aperture=0.001, focus_distance=5.0):
    """The 96-byte camera uniform; defaults are the reference's setupCamera
    (src/renderer/renderer.ts:136-150)."""
    c = np.zeros((), CAMERA)
    c["position"], c["forward"], c["right"], c["up"] = position, forward, right, up
    c["fov"] = fov
    c["aspect"] = (width / height) if aspect is None else aspect
    c["width"], c["height"], c["frame_index"] = width, height, frame_index
    c["aperture"], c["focus_distance"] = aperture, focus_distance
    return c
