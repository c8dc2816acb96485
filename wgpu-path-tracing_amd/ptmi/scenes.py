"""Procedural scenes in the reference's buffer layout.

The reference's assets are GPL-3.0 and are not copied: geometry is regenerated
from dimensions. Each generator returns a `Scene` holding the four storage
blobs of the compute pass (triangles, materials, bvhNodes, lights; reference:
src/shader/pt.wgsl:104-110) plus an optional rgba16float atlas.

Scenes
  cornell()           synthetic Cornell box, 996 triangles, 7 materials, 2 emissive lights
  cornell_spheres()   + three textured PBR spheres sampling a procedural 1024^2 atlas
  grid_1m()           Cornell shell + 708x708 displaced height-field (999 698 + 12 triangles)
  feature_box()       small scene touching every branch of the shader (glass, metal,
                      textures incl. a normal map, point + directional + emissive lights)
"""
from dataclasses import dataclass, field

import numpy as np

from . import layout, scene_host


@dataclass
class Scene:
    name: str
    tris: np.ndarray
    mats: np.ndarray
    nodes: np.ndarray
    lights: np.ndarray
    atlas: np.ndarray = None            # (H, W, 4) float16, or None
    bvh_depth: int = 0
    info: dict = field(default_factory=dict)

    @property
    def scene_bytes(self):
        return self.tris.nbytes + self.nodes.nbytes + self.mats.nbytes + self.lights.nbytes


# ------------------------------------------------------------------ helpers --
def _hash01(i, j, seed):
    """Fixed integer hash -> [0,1) float64 (uint32 arithmetic, wraps)."""
    i = np.asarray(i, np.uint32)
    j = np.asarray(j, np.uint32)
    with np.errstate(over="ignore"):
        h = (i * np.uint32(73856093)) ^ (j * np.uint32(19349663)) ^ np.uint32((seed * 83492791) & 0xFFFFFFFF)
        h ^= h >> np.uint32(16)
        h = h * np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h = h * np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    return h.astype(np.float64) / 4294967296.0


def _material(base=(0.8, 0.8, 0.8), metallic=0.0, roughness=0.5, emission=(0, 0, 0), strength=1.0,
              ior=1.5, transmission=0.0, albedo_map=None, normal_map=None, pbr_map=None, emissive_map=None):
    m = np.zeros((), layout.MATERIAL)
    m["base_color"], m["metallic"], m["roughness"] = base, metallic, roughness
    m["emission"], m["emissive_strength"], m["ior"], m["transmission"] = emission, strength, ior, transmission
    for k, r in (("albedo_map", albedo_map), ("normal_map", normal_map), ("pbr_map", pbr_map),
                 ("emissive_map", emissive_map)):
        if r is not None:
            m[k] = tuple(r)
    return m


def _tri_array(v, n, uv, mat):
    """v, n: (T,3,3); uv: (T,3,2) -> TRIANGLE array."""
    t = np.zeros(len(v), layout.TRIANGLE)
    t["v0"], t["v1"], t["v2"] = v[:, 0], v[:, 1], v[:, 2]
    t["n0"], t["n1"], t["n2"] = n[:, 0], n[:, 1], n[:, 2]
    t["uv0"], t["uv1"], t["uv2"] = uv[:, 0], uv[:, 1], uv[:, 2]
    t["material_index"] = mat
    return t


def _quad(p0, p1, p2, p3, normal, mat):
    """Two triangles, wound so the geometric normal cross(e1,e2) points along `normal`."""
    p = np.array([p0, p1, p2, p3], np.float64)
    nrm = np.array(normal, np.float64)
    if np.dot(np.cross(p[1] - p[0], p[2] - p[0]), nrm) < 0:
        p = p[[0, 3, 2, 1]]
    uvq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64)
    v = np.array([[p[0], p[1], p[2]], [p[0], p[2], p[3]]], np.float32)
    uv = np.array([[uvq[0], uvq[1], uvq[2]], [uvq[0], uvq[2], uvq[3]]], np.float32)
    n = np.broadcast_to(nrm.astype(np.float32), (2, 3, 3)).copy()
    return _tri_array(v, n, uv, mat)


def _box(center, size, mat):
    c = np.array(center, np.float64)
    h = np.array(size, np.float64) / 2
    lo, hi = c - h, c + h
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    q = [
        _quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), (-1, 0, 0), mat),
        _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (1, 0, 0), mat),
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (0, -1, 0), mat),
        _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, 1, 0), mat),
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, -1), mat),
        _quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), (0, 0, 1), mat),
    ]
    return np.concatenate(q)


def _uv_sphere(center, radius, mat, segments=32, rings=16):
    """segments x rings UV sphere: 2*segments cap triangles + 2*segments*(rings-2)
    body triangles (32x16 -> 960), smooth normals, outward winding."""
    c = np.array(center, np.float64)
    tris_v, tris_n, tris_uv = [], [], []

    def pt(i, j):
        th = np.pi * j / rings
        ph = 2 * np.pi * i / segments
        d = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
        return c + radius * d, d, np.array([i / segments, j / rings])

    for j in range(rings):
        for i in range(segments):
            a, b, cc, dd = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            quads = []
            if j == 0:
                quads = [(a, cc, dd)]
            elif j == rings - 1:
                quads = [(a, b, dd)]
            else:
                quads = [(a, b, cc), (a, cc, dd)]
            for tri in quads:
                p = np.array([t[0] for t in tri])
                g = np.cross(p[1] - p[0], p[2] - p[0])
                mid = p.mean(axis=0) - c
                if np.dot(g, mid) < 0:
                    tri = (tri[0], tri[2], tri[1])
                tris_v.append([t[0] for t in tri])
                tris_n.append([t[1] for t in tri])
                tris_uv.append([t[2] for t in tri])
    return _tri_array(np.array(tris_v, np.float32), np.array(tris_n, np.float32),
                      np.array(tris_uv, np.float32), mat)


def _finish(name, tri_parts, mats, punctual=None, atlas=None, info=None):
    tris = np.ascontiguousarray(np.concatenate(tri_parts))
    mats = np.array(mats, layout.MATERIAL)
    nodes, depth = scene_host.build_bvh(tris)            # sorts tris in place (bvh.ts:100-102)
    lights = scene_host.emissive_lights(tris, mats, punctual)
    return Scene(name, tris, mats, nodes, lights, atlas, depth, info or {})


# Cornell dimensions measured from the reference's cornell2.glb
_ROOM_X, _ROOM_Y, _ROOM_Z = 1.0, 2.0015, 1.0
_WHITE = (0.8, 0.8, 0.8)
_RED = (0.8003, 0.0, 0.0618)
_GREEN = (0.0, 0.8005, 0.0540)


def _cornell_shell(m_white, m_red, m_green, m_light, floor=True):
    X, Y, Z = _ROOM_X, _ROOM_Y, _ROOM_Z
    parts = []
    if floor:
        parts.append(_quad((-X, 0, -Z), (X, 0, -Z), (X, 0, Z), (-X, 0, Z), (0, 1, 0), m_white))
    parts += [
        _quad((-X, Y, -Z), (X, Y, -Z), (X, Y, Z), (-X, Y, Z), (0, -1, 0), m_white),      # ceiling
        _quad((-X, 0, -Z), (X, 0, -Z), (X, Y, -Z), (-X, Y, -Z), (0, 0, 1), m_white),     # back wall
        _quad((X, 0, -Z), (X, 0, Z), (X, Y, Z), (X, Y, -Z), (-1, 0, 0), m_red),          # +x wall
        _quad((-X, 0, -Z), (-X, 0, Z), (-X, Y, Z), (-X, Y, -Z), (1, 0, 0), m_green),     # -x wall
    ]
    L, ly = 0.2458, 1.9767
    parts.append(_quad((-L, ly, -L), (L, ly, -L), (L, ly, L), (-L, ly, L), (0, -1, 0), m_light))
    return parts


def cornell(glass=False):
    """SURVEY.md §8d synthetic Cornell: 36 + 960 = 996 triangles, one material per
    primitive (7), two emissive-triangle lights. glass=True turns the tall box
    into a transmissive dielectric (exercises pt.wgsl:522-545, :581-594)."""
    mats = [
        _material(_WHITE), _material(_RED), _material(_GREEN),
        _material(_WHITE, emission=(1, 1, 1), strength=13.8),
        _material((0.887, 1.0, 0.914), roughness=0.5 if not glass else 0.05,
                  transmission=1.0 if glass else 0.0),
        _material(_WHITE, metallic=1.0, roughness=0.0477),
        _material(_WHITE),
    ]
    room = _cornell_shell(0, 1, 2, 3)
    parts = room[:3]                                  # floor, ceiling, back: primitive 0
    parts += [room[3], room[4], room[5]]              # red, green, light
    parts.append(_box((-0.453, 0.288, 0.0), (0.754, 0.585, 0.754), 4))
    parts.append(_box((0.519, 0.213, -0.034), (0.602, 0.426, 0.602), 5))
    parts.append(_uv_sphere((-0.436, 0.776, 0.022), 0.19, 6))
    return _finish("cornell_glass" if glass else "cornell", parts, mats)


def _procedural_atlas(size=1024, tile=256, n_sets=3):
    """rgba16float atlas; set s occupies columns [s*tile, (s+1)*tile), rows:
    0 albedo (checker x gradient), 1 metallic-roughness (g = rough, b = metal),
    2 normal map (hash-perturbed). Returns (atlas, rects[s] = dict)."""
    a = np.zeros((size, size, 4), np.float32)
    a[..., 3] = 1.0
    yy, xx = np.mgrid[0:tile, 0:tile]
    rects = []
    for s in range(n_sets):
        x0 = s * tile
        chk = (((xx // 32) + (yy // 32)) & 1).astype(np.float32)
        grad = (xx / (tile - 1)).astype(np.float32)
        base = np.array([(0.9, 0.3, 0.2), (0.2, 0.6, 0.9), (0.85, 0.8, 0.3)][s % 3], np.float32)
        alb = base[None, None, :] * (0.35 + 0.65 * chk[..., None]) * (0.6 + 0.4 * grad[..., None])
        a[0:tile, x0:x0 + tile, :3] = alb
        rough = 0.15 + 0.7 * _hash01(xx // 16, yy // 16, 11 + s).astype(np.float32)
        metal = (_hash01(xx // 64, yy // 64, 23 + s) > 0.5).astype(np.float32)
        a[tile:2 * tile, x0:x0 + tile, 0] = 1.0
        a[tile:2 * tile, x0:x0 + tile, 1] = rough
        a[tile:2 * tile, x0:x0 + tile, 2] = metal
        nx = (_hash01(xx // 8, yy // 8, 37 + s) - 0.5) * 0.5
        ny = (_hash01(xx // 8, yy // 8, 41 + s) - 0.5) * 0.5
        nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
        a[2 * tile:3 * tile, x0:x0 + tile, 0] = nx * 0.5 + 0.5
        a[2 * tile:3 * tile, x0:x0 + tile, 1] = ny * 0.5 + 0.5
        a[2 * tile:3 * tile, x0:x0 + tile, 2] = nz * 0.5 + 0.5
        rects.append(dict(albedo=(x0, 0, tile, tile), pbr=(x0, tile, tile, tile),
                          normal=(x0, 2 * tile, tile, tile)))
    return a.astype(np.float16), rects


def cornell_spheres():
    """Config 3: Cornell + three radius-0.25 UV spheres (960 triangles each) whose
    materials sample albedo, metallic-roughness and normal rects of the atlas."""
    atlas, rects = _procedural_atlas()
    mats = [
        _material(_WHITE), _material(_RED), _material(_GREEN),
        _material(_WHITE, emission=(1, 1, 1), strength=13.8),
        _material((0.887, 1.0, 0.914)), _material(_WHITE, metallic=1.0, roughness=0.0477), _material(_WHITE),
    ]
    room = _cornell_shell(0, 1, 2, 3)
    parts = room[:3] + [room[3], room[4], room[5]]
    parts.append(_box((-0.453, 0.288, 0.0), (0.754, 0.585, 0.754), 4))
    parts.append(_box((0.519, 0.213, -0.034), (0.602, 0.426, 0.602), 5))
    parts.append(_uv_sphere((-0.436, 0.776, 0.022), 0.19, 6))
    for s, cx in enumerate((-0.6, 0.0, 0.6)):
        r = rects[s]
        mats.append(_material((1, 1, 1), metallic=1.0, roughness=1.0, albedo_map=r["albedo"],
                              pbr_map=r["pbr"], normal_map=r["normal"]))
        parts.append(_uv_sphere((cx, 0.25, 0.68), 0.25, len(mats) - 1))
    return _finish("cornell_spheres", parts, mats, atlas=atlas)


def grid_1m(n=708, amplitude=0.04, seed=1):
    """Config 4: Cornell shell (no flat floor) + an n x n vertex height-field floor
    (2*(n-1)^2 triangles; 708 -> 999 698), heights from the integer hash."""
    mats = [_material(_WHITE), _material(_RED), _material(_GREEN),
            _material(_WHITE, emission=(1, 1, 1), strength=13.8), _material((0.75, 0.75, 0.8), roughness=0.6)]
    parts = _cornell_shell(0, 1, 2, 3, floor=False)
    ii, jj = np.mgrid[0:n, 0:n]
    x = -1.0 + 2.0 * ii / (n - 1)
    z = -1.0 + 2.0 * jj / (n - 1)
    h = amplitude * (0.6 * _hash01(ii // 16, jj // 16, seed) + 0.3 * _hash01(ii // 4, jj // 4, seed + 1)
                     + 0.1 * _hash01(ii, jj, seed + 2))
    P = np.stack([x, h, z], axis=-1)
    gx = np.gradient(h, axis=0) * (n - 1) / 2.0
    gz = np.gradient(h, axis=1) * (n - 1) / 2.0
    N = np.stack([-gx, np.ones_like(h), -gz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    UV = np.stack([ii / (n - 1), jj / (n - 1)], axis=-1)

    def cell(a):
        return a[:-1, :-1], a[1:, :-1], a[1:, 1:], a[:-1, 1:]

    p00, p10, p11, p01 = cell(P)
    n00, n10, n11, n01 = cell(N)
    u00, u10, u11, u01 = cell(UV)
    # wound so the geometric normal points +y: (p00, p01, p11), (p00, p11, p10)
    v = np.concatenate([np.stack([p00, p01, p11], axis=2).reshape(-1, 3, 3),
                        np.stack([p00, p11, p10], axis=2).reshape(-1, 3, 3)]).astype(np.float32)
    nn = np.concatenate([np.stack([n00, n01, n11], axis=2).reshape(-1, 3, 3),
                         np.stack([n00, n11, n10], axis=2).reshape(-1, 3, 3)]).astype(np.float32)
    uv = np.concatenate([np.stack([u00, u01, u11], axis=2).reshape(-1, 3, 2),
                         np.stack([u00, u11, u10], axis=2).reshape(-1, 3, 2)]).astype(np.float32)
    parts.append(_tri_array(v, nn, uv, 4))
    return _finish(f"grid_{n}", parts, mats)


def feature_box():
    """Small scene (few hundred triangles) that reaches every shader branch:
    diffuse, rough metal, glass (front and back faces), a normal-mapped textured
    sphere, an emissive-map-free light, plus a point and a directional light."""
    atlas, rects = _procedural_atlas(size=256, tile=64, n_sets=1)
    r = rects[0]
    mats = [
        _material(_WHITE), _material(_RED), _material(_GREEN),
        _material(_WHITE, emission=(1, 0.9, 0.8), strength=9.0),
        _material((0.9, 1.0, 0.95), roughness=0.08, transmission=1.0, ior=1.45),
        _material((0.9, 0.7, 0.3), metallic=1.0, roughness=0.25),
        _material((1, 1, 1), metallic=1.0, roughness=1.0, albedo_map=r["albedo"], pbr_map=r["pbr"],
                  normal_map=r["normal"]),
        _material((0.6, 0.6, 0.9), metallic=0.4, roughness=0.35),
    ]
    room = _cornell_shell(0, 1, 2, 3)
    parts = room[:3] + [room[3], room[4], room[5]]
    parts.append(_box((-0.45, 0.3, -0.1), (0.5, 0.6, 0.5), 4))
    parts.append(_box((0.5, 0.2, -0.2), (0.5, 0.4, 0.5), 5))
    parts.append(_uv_sphere((0.1, 0.3, 0.5), 0.3, 6, segments=12, rings=8))
    parts.append(_uv_sphere((-0.5, 0.85, -0.1), 0.22, 7, segments=10, rings=6))
    punctual = np.zeros(2, layout.LIGHT)
    punctual[0]["position"], punctual[0]["light_type"] = (0.6, 1.5, 0.7), layout.LIGHT_POINT
    punctual[0]["color"], punctual[0]["intensity"] = (1.0, 0.8, 0.6), 2.0
    punctual[1]["position"], punctual[1]["light_type"] = (0.3, -1.0, -0.4), layout.LIGHT_DIRECTIONAL
    punctual[1]["color"], punctual[1]["intensity"] = (0.6, 0.7, 1.0), 1.5
    return _finish("feature_box", parts, mats, punctual=punctual, atlas=atlas)


def random_soup(seed, n_tris=400):
    """Seeded random scene for fuzzing the parity of the whole path: triangles of mixed sizes inside [-1,1] x [0,2] x
    [-1,1] with un-normalised, sometimes flipped or zero vertex normals, a few degenerate triangles (zero area,
    repeated vertices), random materials (every lobe, some textured), one emissive material, and a point plus an
    axis-aligned directional light (its shadow rays have zero direction components)."""
    rng = np.random.default_rng(seed)
    atlas, rects = _procedural_atlas(size=128, tile=32, n_sets=1)
    r = rects[0]
    mats = [_material(_WHITE, emission=(1, 0.95, 0.9), strength=6.0)]
    for k in range(9):
        lobe = k % 3
        mats.append(_material(tuple(rng.random(3) * 0.8 + 0.1),
                              metallic=float(rng.random()) if lobe == 1 else 0.0,
                              roughness=float(rng.choice([0.0, 0.02, 0.3, 1.0])),
                              transmission=float(rng.choice([1.0, 0.5])) if lobe == 2 else 0.0,
                              ior=float(rng.choice([1.0, 1.33, 1.5, 2.4])),
                              albedo_map=r["albedo"] if k in (3, 4) else None,
                              pbr_map=r["pbr"] if k == 4 else None, normal_map=r["normal"] if k in (4, 5) else None))
    c = rng.random((n_tris, 1, 3)) * np.array([2.0, 2.0, 2.0]) + np.array([-1.0, 0.0, -1.0])
    size = rng.choice([0.02, 0.1, 0.4, 1.5], size=(n_tris, 1, 1), p=[0.2, 0.5, 0.25, 0.05])
    v = (c + (rng.random((n_tris, 3, 3)) - 0.5) * size).astype(np.float32)
    v[0::37, 1] = v[0::37, 0]                                   # repeated vertex
    v[5::41, 2] = (v[5::41, 0] + v[5::41, 1]) / 2               # collinear
    geo = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    n = np.repeat(geo[:, None, :], 3, 1) + (rng.random((n_tris, 3, 3)) - 0.5) * 0.3 * np.abs(geo).max()
    n[3::29] *= -1.0                                            # shading normal opposite to the geometric one
    n[7::53] = 0.0                                              # zero normals: normalize() gives NaN, as in the reference
    uv = (rng.random((n_tris, 3, 2)) * 3 - 1).astype(np.float32)
    mat = rng.integers(1, len(mats), n_tris)
    mat[:6] = 0                                                 # six emissive triangles
    tris = _tri_array(v, n.astype(np.float32), uv, mat)
    floor = _quad((-1.2, -0.01, -1.2), (1.2, -0.01, -1.2), (1.2, -0.01, 1.2), (-1.2, -0.01, 1.2), (0, 1, 0), 1)
    punctual = np.zeros(2, layout.LIGHT)
    punctual[0]["position"], punctual[0]["light_type"] = tuple(rng.random(3) * [1.6, 0.5, 1.6] + [-0.8, 1.6, -0.8]), layout.LIGHT_POINT
    punctual[0]["color"], punctual[0]["intensity"] = (1.0, 0.9, 0.8), 3.0
    punctual[1]["position"], punctual[1]["light_type"] = (0.0, -1.0, 0.0), layout.LIGHT_DIRECTIONAL
    punctual[1]["color"], punctual[1]["intensity"] = (0.7, 0.8, 1.0), 1.0
    return _finish("random_soup_%d" % seed, [tris, floor], mats, punctual=punctual, atlas=atlas)


def deep_chain(n=1000, ratio=1.08):
    """Triangles whose position and size grow geometrically along +x: the reference builder always cuts the biggest
    twelfth off, so the BVH is a 35-level chain (root = 1) — deeper than any fixed on-chip stack this library uses —
    and a ray along the chain keeps one pending far child per level."""
    k = np.arange(n)
    x = (ratio ** k - 1.0) * 1e-12
    s = 0.5 * ratio ** k * 1e-12
    v = np.zeros((n, 3, 3), np.float64)
    v[:, 0] = np.stack([x, 0 * x, 0 * x], 1)
    v[:, 1] = np.stack([x + s, 0 * x, 0.3 * s], 1)
    v[:, 2] = np.stack([x, s, 0 * x], 1)
    nrm = np.zeros((n, 3, 3), np.float32)
    nrm[..., 2] = 1.0
    tris = _tri_array(v.astype(np.float32), nrm, np.zeros((n, 3, 2), np.float32), np.arange(n) % 2)
    mats = [_material(_WHITE), _material(_WHITE, emission=(1, 1, 1), strength=4.0)]
    return _finish("deep_chain", [tris], mats)


def cornell_enclosed():
    """The Cornell box inside a closed white room that also holds the camera: no camera ray leaves the scene, so every path
    survives bounce 0 and the bounce-1 queue is dense (a probe for how much the kernels lose to sparse path ids)."""
    base = cornell()
    room = _box((0.0, 2.0, 1.5), (8.0, 6.0, 9.0), 0)
    for k in ("n0", "n1", "n2"):
        room[k] = -room[k]                      # the room is seen from inside
    return _finish("cornell_enclosed", [base.tris, room], base.mats)


SCENES = {"cornell": cornell, "cornell_glass": lambda: cornell(glass=True), "cornell_enclosed": cornell_enclosed,
          "cornell_spheres": cornell_spheres, "grid_1m": grid_1m, "feature_box": feature_box, "deep_chain": deep_chain}


def make(name):
    return SCENES[name]()
