#!/usr/bin/env python3
"""tools/multi_gather_time.py: what the multi-device handle (ptmi_multi_*, include/ptmi.h) costs on a one-GPU box.
  * N = 1 through RCCL (ncclCommInitAll over one device): BASELINE configs[4]'s 3840x2160 frame, one 64-frame dispatch, then the
    gather (pack kernel -> ncclGather -> unpack kernel) timed by the library's own events (ptmi_multi_gather_ms), and the same
    dispatch on a plain context beside it — the handle must not cost throughput;
  * N = 8 contexts on the one device (loopback copies in place of the collective): the pack / copy / unpack of 8 x 270 rows.
N > 1 over RCCL / xGMI cannot run here (one GPU per box)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
from ptmi import layout, native, scenes  # noqa: E402

W, H, F = 3840, 2160, 64
sc = scenes.make("cornell")
cam_kw = dict(aperture=0.05, focus_distance=2.8)
out = {"frame": [W, H], "frames_per_dispatch": F}


def timed(obj, frame0):
    obj.synchronize()
    t = time.time()
    obj.dispatch(layout.make_camera(W, H, frame_index=frame0, **cam_kw), F)
    obj.synchronize()
    return time.time() - t


with native.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.resize(W, H); ctx.set_options(max_bounces=8, do_mis=1)
    timed(ctx, 0)
    ctx.reset_stats()
    dt = timed(ctx, F)
    out["plain_context"] = {"dispatch_s": round(dt, 4), "msamples_per_s": round(ctx.stats().segments / dt / 1e6, 1)}

with native.MultiContext([0]) as m:
    m.upload_scene(sc); m.resize(W, H); m.set_options(max_bounces=8, do_mis=1)
    timed(m, 0); m.gather(); m.synchronize()
    m.reset_stats()
    dt = timed(m, F)
    t = time.time(); m.gather(); m.synchronize(); wall = time.time() - t
    out["multi_n1_rccl"] = {"dispatch_s": round(dt, 4), "msamples_per_s": round(m.stats().segments / dt / 1e6, 1),
                            "gather_ms_events": round(m.gather_ms(), 3), "gather_ms_wall": round(wall * 1e3, 3),
                            "bytes_gathered": W * H * 16}

with native.MultiContext([0] * 8, loopback=True) as m:
    m.upload_scene(sc); m.resize(W, H); m.set_options(max_bounces=8, do_mis=1, frames_per_batch=8)
    m.dispatch(layout.make_camera(W, H, **cam_kw), 8); m.gather(); m.synchronize()
    t = time.time(); m.gather(); m.synchronize(); wall = time.time() - t
    o = m.options()
    out["multi_n8_loopback_one_device"] = {"tile_strip": int(o.tile_strip), "gather_ms_events": round(m.gather_ms(), 3),
                                           "gather_ms_wall": round(wall * 1e3, 3),
                                           "note": "8 contexts share one GPU: only the packing / copies / unpacking are meaningful, not the render time"}
# RCCL prints a version banner on stdout when it is loaded: the JSON goes to the file named on the command line when there is one
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
