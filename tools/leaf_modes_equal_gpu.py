#!/usr/bin/env python3
"""BASELINE.json's GPU configurations at their FULL sample counts, rendered over the library's own leaves (ptmi_options.leaves = 2) and over
the reference's (leaves = 1, the mode the parity tests pin to the oracle): the frames must be equal bit for bit and the counters equal —
every ray of the render found the same (t, triangle) / verdict, or a pixel would differ. Prints the rays counted (path segments + traced
shadow rays) and how many were traced again after a failed verification.
usage (GPU box): python tools/leaf_modes_equal_gpu.py [repeat=1] [leaves|cull]    (repeat: more accumulation passes with later frame indices)
With `cull`: the distance cull (ptmi_options.cull = 1, the default: boxes beyond the closest hit so far are skipped) against
cull = 0 over the reference's leaves — the two may differ where Moller-Trumbore reports a hit outside its triangle's box (a grazing ray);
the run counts how often."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
import numpy as np
from ptmi import layout, native, scenes

repeat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
what = sys.argv[2] if len(sys.argv) > 2 else "leaves"
CASES = [("configs[1]", "cornell", 1920, 1080, 64, 0.001, 5.0), ("configs[2]", "cornell_spheres", 1920, 1080, 512, 0.001, 5.0),
         ("configs[3]", "grid_1m", 1920, 1080, 64, 0.001, 5.0), ("configs[4]", "cornell", 3840, 2160, 256, 0.05, 2.8)]
ctx = native.Context(0)
total, bad = 0, 0
for tag, name, W, H, spp, ap, focus in CASES:
    sc = scenes.grid_1m() if name == "grid_1m" else scenes.make(name)
    for rep in range(repeat):
        imgs, stats = [], []
        for mode in (1, 2):
            leaves, cull = (mode, 1) if what == "leaves" else (1, 2 - mode)
            ctx.set_options(leaves=leaves, leaf_tris=0, keep_reference_tree=0, traversal=native.TRAVERSAL_AUTO, cull=cull)
            ctx.upload_scene(sc); ctx.resize(W, H)
            ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, frames_per_batch=0, cull=cull)
            ctx.reset_stats()
            cam = layout.make_camera(W, H, aperture=ap, focus_distance=focus, frame_index=rep * spp)
            t0 = time.time()
            for f0 in range(0, spp, 64):
                cam["frame_index"] = rep * spp + f0
                ctx.dispatch(cam, min(64, spp - f0))
            imgs.append(ctx.read_output()); stats.append(ctx.stats())
        a, b = stats
        rays = b.segments + b.shadow_traced
        same_counts = (a.segments, a.shadow_rays, a.shadow_traced) == (b.segments, b.shadow_rays, b.shadow_traced)
        x, y = imgs[0].view(np.uint32), imgs[1].view(np.uint32)
        diff = int(((x != y) & ~(np.isnan(imgs[0]) & np.isnan(imgs[1]))).sum())
        total += rays; bad += diff + (0 if same_counts else 1)
        print(f"{tag} {name} {W}x{H} x {spp} spp (frames {rep * spp}..): {rays} rays, counters equal {same_counts}, differing floats {diff}, "
              f"traced again {b.verify_failed}, variants {b.extend_variant}/{b.shadow_variant}", flush=True)
print(f"total {total} rays, {bad} mismatches")
sys.exit(1 if bad else 0)
