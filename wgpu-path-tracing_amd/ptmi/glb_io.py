"""Tiny binary-glTF writer (and blob loader) used to feed the Node host's .glb path in tests and demos.
Only what the reference's assets use: float32 POSITION / NORMAL / TEXCOORD_0, uint16 indices, TRS node
hierarchies, pbrMetallicRoughness + emissive / transmission / ior extensions, KHR_lights_punctual."""
import json
import struct

import numpy as np

from . import layout, scenes


def write_glb(path, meshes, nodes, materials, lights=None):
    """meshes: list of dicts {positions (N,3), normals (N,3), uvs (N,2) or None, indices (M,), material or None};
    nodes: list of dicts {mesh?, light?, translation?, rotation?, scale?, matrix?, children?}."""
    bin_parts, views, accessors = [], [], []

    def add(arr, target, ctype, atype):
        data = np.ascontiguousarray(arr).tobytes()
        off = sum(len(b) for b in bin_parts)
        bin_parts.append(data + b"\0" * (-len(data) % 4))
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(data), "target": target})
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(len(arr)), "type": atype}
        if atype == "VEC3":
            a = np.asarray(arr, np.float64)
            acc["min"], acc["max"] = a.min(0).tolist(), a.max(0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    jm = []
    for m in meshes:
        attrs = {"POSITION": add(np.asarray(m["positions"], np.float32), 34962, 5126, "VEC3"),
                 "NORMAL": add(np.asarray(m["normals"], np.float32), 34962, 5126, "VEC3")}
        if m.get("uvs") is not None:
            attrs["TEXCOORD_0"] = add(np.asarray(m["uvs"], np.float32), 34962, 5126, "VEC2")
        prim = {"attributes": attrs, "indices": add(np.asarray(m["indices"], np.uint16), 34963, 5123, "SCALAR")}
        if m.get("material") is not None:
            prim["material"] = m["material"]
        jm.append({"primitives": [prim]})
    jn = []
    for n in nodes:
        o = {k: n[k] for k in ("mesh", "translation", "rotation", "scale", "matrix", "children", "name") if k in n}
        if "light" in n:
            o["extensions"] = {"KHR_lights_punctual": {"light": n["light"]}}
        jn.append(o)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": jn,
           "meshes": jm, "materials": materials, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": sum(len(b) for b in bin_parts)}]}
    if lights:
        doc["extensions"] = {"KHR_lights_punctual": {"lights": lights}}
        doc["extensionsUsed"] = ["KHR_lights_punctual"]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    binary = b"".join(bin_parts)
    total = 12 + 8 + len(js) + 8 + len(binary)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(binary), 0x004E4942) + binary)


def load_blob_dir(d, name="glb"):
    """The .bin blobs written by host/prepare_cli.js -> Scene."""
    import os
    rd = lambda n, dt: np.fromfile(os.path.join(d, n + ".bin"), dt)
    return scenes.Scene(name, rd("triangles", layout.TRIANGLE), rd("materials", layout.MATERIAL),
                        rd("bvhNodes", layout.BVH_NODE), rd("lights", layout.LIGHT), None)
