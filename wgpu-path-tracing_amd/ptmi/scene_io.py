"""Scene interchange with the Node host: the .ptscene container read by host/scene_file.js and a
SceneData JSON in the shape of the reference's CPU types (src/renderer/gpu.ts:10-65)."""
import json
import struct

import numpy as np

from . import layout


def save_ptscene(scene, path):
    """"PTSC" | u32 version | u32 jsonLength | json | blobs (16-byte aligned)."""
    parts = [("triangles", scene.tris.tobytes()), ("materials", scene.mats.tobytes()),
             ("bvhNodes", scene.nodes.tobytes()), ("lights", scene.lights.tobytes())]
    atlas = scene.atlas
    if atlas is not None:
        parts.append(("atlas", np.ascontiguousarray(atlas).tobytes()))
    meta = {k: {"offset": 0, "length": len(b)} for k, b in parts}
    if atlas is not None:
        meta["atlas"].update(width=int(atlas.shape[1]), height=int(atlas.shape[0]),
                             format=1 if atlas.dtype == np.float16 else 2)
    for _ in range(2):                                     # offsets depend on the json length: settle in two passes
        js = json.dumps(meta).encode()
        js += b" " * (-(12 + len(js)) % 16)
        off = 12 + len(js)
        for k, b in parts:
            meta[k]["offset"] = off
            off += len(b) + (-len(b) % 16)
        js2 = json.dumps(meta).encode()
        js2 += b" " * (-(12 + len(js2)) % 16)
        if len(js2) == len(js):
            js = js2
            break
    with open(path, "wb") as f:
        f.write(b"PTSC" + struct.pack("<II", 1, len(js)) + js)
        for k, b in parts:
            assert f.tell() == meta[k]["offset"]
            f.write(b + b"\0" * (-len(b) % 16))


def _f(a):
    return [float(x) for x in a]


def scene_data_json(scene, camera):
    """{scene: SceneData, camera: CameraCPU} as plain JSON numbers (f32 values are exact in f64)."""
    rect = lambda r: {"x": int(r["x"]), "y": int(r["y"]), "w": int(r["w"]), "h": int(r["h"])}
    tris = [dict(v0=_f(t["v0"]), v1=_f(t["v1"]), v2=_f(t["v2"]), n0=_f(t["n0"]), n1=_f(t["n1"]), n2=_f(t["n2"]),
                 uv0=_f(t["uv0"]), uv1=_f(t["uv1"]), uv2=_f(t["uv2"]), materialIndex=int(t["material_index"]))
            for t in scene.tris]
    mats = [dict(baseColor=_f(m["base_color"]), metallic=float(m["metallic"]), roughness=float(m["roughness"]),
                 emission=_f(m["emission"]), emissiveStrength=float(m["emissive_strength"]), ior=float(m["ior"]),
                 transmission=float(m["transmission"]), albedoMap=rect(m["albedo_map"]), normalMap=rect(m["normal_map"]),
                 pbrMap=rect(m["pbr_map"]), emissiveMap=rect(m["emissive_map"])) for m in scene.mats]
    # leaves carry -1 in the reference's JS objects (bvh.ts:87-88); the packer turns it into 0xFFFFFFFF
    s32 = lambda v: -1 if int(v) == 0xFFFFFFFF else int(v)
    nodes = [dict(aabb=dict(min=_f(n["aabb_min"]), max=_f(n["aabb_max"])), left=s32(n["left"]), right=s32(n["right"]),
                  triangleOffset=int(n["triangle_offset"]), triangleCount=int(n["triangle_count"])) for n in scene.nodes]
    lights = [dict(position=_f(l["position"]), lightType=int(l["light_type"]), color=_f(l["color"]),
                   intensity=float(l["intensity"]), triangleIndex=int(l["triangle_index"])) for l in scene.lights]
    c = camera
    cam = dict(position=_f(c["position"]), forward=_f(c["forward"]), right=_f(c["right"]), up=_f(c["up"]),
               fov=float(c["fov"]), aspect=float(c["aspect"]), width=int(c["width"]), height=int(c["height"]),
               frameIndex=int(c["frame_index"]), aperture=float(c["aperture"]), focusDistance=float(c["focus_distance"]))
    return {"scene": dict(triangles=tris, materials=mats, bvhNodes=nodes, lights=lights), "camera": cam}
