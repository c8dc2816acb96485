#!/usr/bin/env python3
"""tools/multi_vs_plain.py: does the multi-device handle cost throughput at N = 1? Both handles alive at once, five alternating 64-frame
dispatches of BASELINE configs[4]'s frame each (after a warm-up each), Msamples/s per dispatch."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
from ptmi import layout, native, scenes  # noqa: E402

W, H, F = 3840, 2160, 64
sc = scenes.make("cornell")
kw = dict(aperture=0.05, focus_distance=2.8)
ctx = native.Context(0); m = native.MultiContext([0])
for o in (ctx, m):
    o.upload_scene(sc); o.resize(W, H); o.set_options(max_bounces=8, do_mis=1, frames_per_batch=8)   # two handles share the card: small batches


def run(o, f0):
    o.synchronize(); o.reset_stats()
    t = time.time(); o.dispatch(layout.make_camera(W, H, frame_index=f0, **kw), F); o.synchronize(); dt = time.time() - t
    return round(o.stats().segments / dt / 1e6, 1)


run(ctx, 0); run(m, 0)
out = {"plain": [], "multi_n1": []}
for k in range(1, 6):
    out["plain"].append(run(ctx, k * F)); out["multi_n1"].append(run(m, k * F))
out["mean_ratio_multi_over_plain"] = round(sum(out["multi_n1"]) / sum(out["plain"]), 4)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
