#!/usr/bin/env python3
"""Generates the committed fixtures tests/golden/*.npz.

The reference has no golden vectors for the path-tracing path and cannot run here
(SURVEY.md §8c), so these are outputs of this repository's CPU oracle (contract build) on
small inputs — regression pins for the oracle itself (other compilers / CPUs / numpy
versions must reproduce them) and fixed targets for the HIP path. Each file holds the
complete inputs (scene blobs, camera, rays) next to the expected outputs.

    python tests/golden/make_golden.py [fixture names]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_lib import Oracle  # noqa: E402
from ptmi import layout, scenes  # noqa: E402

CASES = [  # name, scene, W, H, frames, bounces, mis, aperture
    ("cornell_64x48_4spp_mis", "cornell", 64, 48, 4, 8, 1, 0.001),
    ("cornell_64x64_4spp_b4_nomis", "cornell", 64, 64, 4, 4, 0, 0.001),      # BASELINE configs[0] shape, reduced
    ("feature_box_48x48_3spp", "feature_box", 48, 48, 3, 8, 1, 0.05),
    ("random_soup3_48x48_3spp", "random_soup:3", 48, 48, 3, 8, 1, 0.02),        # degenerate triangles, zero normals, all lobes
    ("deep_chain_32x32_2spp", "deep_chain", 32, 32, 2, 4, 1, 0.0),              # 58-level BVH (spilling node stacks)
]


def main():
    orc = Oracle(strict=False)
    only = set(sys.argv[1:])                  # optional: names of the fixtures to (re)write; default all
    for name, sname, W, H, frames, bounces, mis, ap in CASES:
        if only and name not in only:
            continue
        sc = scenes.random_soup(int(sname.split(":")[1])) if sname.startswith("random_soup:") else scenes.make(sname)
        cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
        out, st = orc.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
        rng = np.random.default_rng(1234)
        n = 4096
        d = rng.standard_normal((n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        lo, hi = sc.nodes[0]["aabb_min"], sc.nodes[0]["aabb_max"]
        o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
        if sname == "deep_chain":             # rays along the chain: every level of the tree keeps a pending far child
            o = (np.array([-1e-12, 0, 0]) + rng.standard_normal((n, 3)) * 1e-13).astype(np.float32)
            d = np.array([1.0, 0, 0]) + rng.standard_normal((n, 3)) * rng.choice([0, 1e-3, 0.05], (n, 1))
            d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        t, tri, u, v, _ = orc.intersect(sc, o, d)
        dist = (rng.random(n) * 2.0).astype(np.float32)
        dist[::4] = -1
        occ = orc.occluded(sc, o, d, dist)
        atlas = sc.atlas if sc.atlas is not None else np.zeros((0, 0, 4), np.float16)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            tris=sc.tris.view(np.uint8), mats=sc.mats.view(np.uint8), nodes=sc.nodes.view(np.uint8),
            lights=sc.lights.view(np.uint8), atlas=atlas, camera=np.frombuffer(cam.tobytes(), np.uint8),
            frames=frames, bounces=bounces, mis=mis, image=out, segments=st.segments, shadow_rays=st.shadow_rays,
            ray_o=o, ray_d=d, hit_t=t, hit_tri=tri, hit_u=u, hit_v=v, shadow_dist=dist, shadow_occluded=occ)
        print(name, out[..., :3].mean(), st.segments, st.shadow_rays, (t > 0).mean(), occ.mean())


if __name__ == "__main__":
    main()
