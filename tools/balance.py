#!/usr/bin/env python3
"""Load balance of the two row shardings at N ranks (one GPU, parts rendered one after the other):
segments and device time per part, contiguous bands vs interleaved 4-row strips."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
import numpy as np
from ptmi import layout, native, scenes, shard
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = shard.weak_frame(1920, 1080, N)
ctx = native.Context(0); ctx.upload_scene(scenes.make("cornell")); ctx.resize(W, H)
cam = layout.make_camera(W, H)
for name in ("bands", "strips"):
    seg, ms = [], []
    for r in range(N):
        if name == "bands":
            y0, y1 = shard.band(H, N, r); opts = dict(tile_y0=y0, tile_y1=y1, tile_parts=0, tile_part=0)
        else:
            opts = shard.strip_options(N, r)
        ctx.set_options(timing=1, **opts)
        ctx.dispatch(cam, frames); ctx.reset_stats()           # warm-up (allocations)
        ctx.dispatch(cam, frames)
        st = ctx.stats(); seg.append(st.segments); ms.append(st.gpu_ms)
    seg, ms = np.array(seg, float), np.array(ms)
    print(f"{name:7s} N={N} frame {W}x{H}: segments max/mean {seg.max() / seg.mean():.3f}  device ms max/mean {ms.max() / ms.mean():.3f}  "
          f"(ms per part: {' '.join('%.1f' % m for m in ms)})  -> efficiency bound {ms.mean() / ms.max():.3f}")
