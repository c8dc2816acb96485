"""Tiny binary-glTF writer (and blob loader) used to feed the Node host's .glb path in tests and demos.
Only what the reference's assets use: float32 POSITION / NORMAL / TEXCOORD_0, uint16 indices, TRS node
hierarchies, pbrMetallicRoughness + emissive / transmission / ior extensions, KHR_lights_punctual, and
PNG images embedded as buffer views (what textured exports carry; src/renderer/atlas.ts consumes them)."""
import json
import struct
import zlib

import numpy as np

from . import layout, scenes


def _png_chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def encode_png(pixels, color_type=6, filters=None, palette=None, trns=None, depth=8):
    """pixels: (H, W, C) uint8 for colour types 0 (C=1), 2 (3), 4 (2), 6 (4), or (H, W) palette indices for
    type 3 (depth 1/2/4/8 packs the indices / grey levels). filters: one PNG filter type 0..4 per row
    (default: cycles through all five, so a decoder sees every one)."""
    px = np.asarray(pixels, np.uint8)
    h, w = px.shape[:2]
    if depth < 8:
        assert color_type in (0, 3)
        per = 8 // depth
        idx = px.reshape(h, w).astype(np.uint16)
        padded = np.zeros((h, -(-w // per) * per), np.uint16)
        padded[:, :w] = idx
        rows = np.zeros((h, padded.shape[1] // per), np.uint16)
        for k in range(per):
            rows |= padded[:, k::per] << (8 - depth * (k + 1))
        rows = rows.astype(np.uint8)
        bpp = 1
    else:
        rows = px.reshape(h, -1)
        bpp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    out = bytearray()
    zero = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        f = (y % 5) if filters is None else filters[y]
        cur = rows[y].astype(np.int32)
        up = rows[y - 1].astype(np.int32) if y else zero
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        pred = [zero, left, up, (left + up) >> 1, _paeth(left, up, upleft)][f]
        out.append(f)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
    body = _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        body += _png_chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        body += _png_chunk(b"tRNS", bytes(trns))
    raw = zlib.compress(bytes(out))
    half = len(raw) // 2                                   # two IDAT chunks: decoders must concatenate them
    body += _png_chunk(b"IDAT", raw[:half]) + _png_chunk(b"IDAT", raw[half:]) + _png_chunk(b"IEND", b"")
    return b"\x89PNG\r\n\x1a\n" + body


def write_glb(path, meshes, nodes, materials, lights=None, images=None, textures=None):
    """meshes: list of dicts {positions (N,3), normals (N,3), uvs (N,2) or None, indices (M,), material or None};
    nodes: list of dicts {mesh?, light?, translation?, rotation?, scale?, matrix?, children?};
    images: list of encoded image files (bytes, PNG); textures: list of image indices (one glTF texture each)."""
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
    if images:
        doc["images"] = []
        for data in images:
            off = sum(len(b) for b in bin_parts)
            bin_parts.append(data + b"\0" * (-len(data) % 4))
            views.append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
            doc["images"].append({"bufferView": len(views) - 1, "mimeType": "image/png"})
        doc["textures"] = [{"source": int(i)} for i in (textures if textures is not None else range(len(images)))]
        doc["buffers"] = [{"byteLength": sum(len(b) for b in bin_parts)}]
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


def scene_to_glb(scene, path):
    """Writes a Scene (ptmi.scenes) as a binary glTF: one mesh primitive per material, unindexed triangles in the
    scene's order, material factors and the emissive-strength / transmission / ior extensions the reference's loader
    reads (src/renderer/gpu.ts:275-399). Textures are not written (the synthetic Cornell has none). The Node host loads
    the file like any other .glb — parse, world transforms, BVH, light list — so BASELINE configs[0] ("Cornell Box glTF")
    can be rendered literally."""
    tris, mats = scene.tris, scene.mats
    meshes, nodes, materials = [], [], []
    for mi in range(len(mats)):
        t = tris[tris["material_index"] == mi]
        if not len(t):
            continue
        pos = np.stack([t["v0"], t["v1"], t["v2"]], 1).reshape(-1, 3)
        nrm = np.stack([t["n0"], t["n1"], t["n2"]], 1).reshape(-1, 3)
        uv = np.stack([t["uv0"], t["uv1"], t["uv2"]], 1).reshape(-1, 2)
        if len(pos) > 65535:
            raise ValueError("scene_to_glb writes uint16 indices: at most 21845 triangles per material")
        m = mats[mi]
        f = lambda v: [float(x) for x in np.atleast_1d(v)]
        mat = {"pbrMetallicRoughness": {"baseColorFactor": f(m["base_color"]) + [1.0], "metallicFactor": float(m["metallic"]),
                                        "roughnessFactor": float(m["roughness"])},
               "extensions": {"KHR_materials_ior": {"ior": float(m["ior"])},
                              "KHR_materials_transmission": {"transmissionFactor": float(m["transmission"])}}}
        if np.any(m["emission"] > 0):
            mat["emissiveFactor"] = f(m["emission"])
            mat["extensions"]["KHR_materials_emissive_strength"] = {"emissiveStrength": float(m["emissive_strength"])}
        meshes.append({"positions": pos, "normals": nrm, "uvs": uv, "indices": np.arange(len(pos)), "material": len(materials)})
        materials.append(mat)
        nodes.append({"mesh": len(meshes) - 1})
    write_glb(path, meshes, nodes, materials)


def load_blob_dir(d, name="glb"):
    """The .bin blobs written by host/prepare_cli.js -> Scene."""
    import os
    rd = lambda n, dt: np.fromfile(os.path.join(d, n + ".bin"), dt)
    atlas = None
    if os.path.exists(os.path.join(d, "atlas.bin")):
        with open(os.path.join(d, "info.json")) as f:
            a = json.load(f)["atlas"]
        atlas = rd("atlas", np.float16).reshape(a["height"], a["width"], 4)
    return scenes.Scene(name, rd("triangles", layout.TRIANGLE), rd("materials", layout.MATERIAL),
                        rd("bvhNodes", layout.BVH_NODE), rd("lights", layout.LIGHT), atlas)
