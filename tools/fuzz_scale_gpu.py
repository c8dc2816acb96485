#!/usr/bin/env python3
"""One-off: random scenes scaled to extreme sizes (denormal-range and huge coordinates) — the arithmetic contract says
f32 denormals are kept and overflow follows IEEE; GPU and oracle must still agree bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ptmi import layout, native, scenes, scene_host
from oracle_lib import Oracle
oracle, ctx = Oracle(), native.Context(0)
bad = 0
for seed, scale in ((0, 1e-18), (1, 1e-10), (2, 1e-30), (3, 1e8), (4, 1e15), (5, 3e-37), (6, 1e18)):
    base = scenes.random_soup(seed)
    tris = base.tris.copy()
    for k in ("v0", "v1", "v2"):
        tris[k] = (tris[k].astype(np.float64) * scale).astype(np.float32)
    lights = base.lights.copy()
    pt = lights["light_type"] == layout.LIGHT_POINT
    lights["position"][pt] = (lights["position"][pt].astype(np.float64) * scale).astype(np.float32)
    nodes, depth = scene_host.build_bvh(tris)
    em = scene_host.emissive_lights(tris, base.mats, lights[lights["light_type"] != layout.LIGHT_EMISSIVE])
    sc = scenes.Scene("scaled", tris, base.mats, nodes, em, base.atlas, depth)
    W, H, frames = 96, 64, 4
    cam = layout.make_camera(W, H, aperture=0.0, focus_distance=2.5 * scale)
    cam["position"] = (np.array([0, 1.0, 2.8]) * scale).astype(np.float32)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    for trav in (native.TRAVERSAL_AUTO, native.TRAVERSAL_GLOBAL, native.TRAVERSAL_GLOBAL_EXACT):     # GLOBAL: the quantised image
        ctx.upload_scene(sc); ctx.resize(W, H)
        ctx.set_options(max_bounces=8, do_mis=1, frames_per_batch=0, cull=1, traversal=trav, tile_y0=0, tile_y1=0, tile_parts=0)
        ctx.reset_stats(); ctx.dispatch(cam, frames); got = ctx.read_output(); st = ctx.stats()
        diff = (got.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(got) & np.isnan(ref))
        ok = not diff.any() and (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays)
        bad += not ok
        print(f"seed {seed} scale {scale:g} trav {trav} depth {depth} segs {st.segments}/{ost.segments} shadow {st.shadow_rays}/{ost.shadow_rays} "
              f"mean {np.nanmean(ref[..., :3]):.4g} {'OK' if ok else 'MISMATCH %d floats' % diff.sum()}", flush=True)
print("mismatching runs:", bad); sys.exit(1 if bad else 0)
