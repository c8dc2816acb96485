#!/usr/bin/env python3
"""Determinism / stability soak: the same dispatches repeated must give the same bits and counters every time.
usage (GPU box): python tools/soak_gpu.py [rounds=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
import numpy as np
from ptmi import layout, native, scenes
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = native.Context(0)
cases = [("cornell", 1920, 1080, 64, {}), ("cornell_spheres", 1280, 720, 16, {}), ("feature_box", 640, 480, 32, {}),
         ("cornell", 1920, 1080, 16, dict(tile_parts=3, tile_part=1, tile_strip=4)), ("grid_1m", 960, 540, 8, {})]
ref = {}
t0 = time.time(); bad = 0
for r in range(rounds):
    for i, (name, W, H, frames, opts) in enumerate(cases):
        sc = scenes.make(name) if r == 0 else ref[i][2]
        ctx.upload_scene(sc); ctx.resize(W, H)
        base = dict(max_bounces=8, do_mis=1, frames_per_batch=0, tile_y0=0, tile_y1=0, tile_parts=0, tile_part=0, tile_strip=0)
        base.update(opts); ctx.set_options(**base); ctx.reset_stats()
        ctx.dispatch(layout.make_camera(W, H), frames)
        out = ctx.read_output().view(np.uint32).copy(); st = ctx.stats()
        key = (st.segments, st.shadow_rays, st.paths)
        if r == 0:
            ref[i] = (out, key, sc)
        elif not (np.array_equal(out, ref[i][0]) and key == ref[i][1]):
            bad += 1; print("round", r, "case", i, name, "DIFFERS", key, ref[i][1], flush=True)
    if r % 5 == 0:
        print(f"round {r} done, {time.time() - t0:.1f} s", flush=True)
print("rounds", rounds, "differences", bad)
sys.exit(1 if bad else 0)
