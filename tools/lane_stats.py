#!/usr/bin/env python3
"""Where do the traversal kernels' lanes idle?  Runs one dispatch of a BASELINE config on a diagnostic build of the library
(make -C wgpu-path-tracing_amd variant NAME=util EXTRA=-DPT_UTIL_STATS; PTMI_LIB=.../lib/ab/libptmi_util.so) and prints, per
kernel kind, the wave-level step counts and the lanes that took part.   usage: PTMI_LIB=... tools/lane_stats.py [config]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
from ptmi import layout, native, scenes  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = {1: "cornell", 2: "cornell_spheres", 3: "grid_1m", 4: "cornell"}[cfg]
W, H, frames = 1920, 1080, 16
sc = scenes.make(name)
ctx = native.Context(0)
ctx.set_options(max_bounces=8, do_mis=1, **json.loads(os.environ.get("PTMI_OPTS", "{}")))      # (before the upload: `leaves` is read there)
ctx.upload_scene(sc)
ctx.resize(W, H)
# e.g. PTMI_OPTS='{"leaves": 1, "traversal": 2}'
lib = ctypes.CDLL(os.environ["PTMI_LIB"])
out = (ctypes.c_ulonglong * 32)()
ctx.dispatch(layout.make_camera(W, H), frames)
ctx.read_output()
assert lib.ptmi_debug_util_stats(out, 1) == 0
st = ctx.stats()
res = {}
for k, kind in enumerate(("extend", "shadow")):
    u = list(out[16 * k:16 * k + 10])
    votes, held, refills, refilled, nsteps, nlanes, lsteps, llanes, titer, tlanes = u
    rays = st.segments if k == 0 else st.shadow_traced
    res[kind] = {
        "rays": rays,
        "votes_per_ray_x64": round(64 * votes / rays, 3),
        "lanes_holding_a_ray_at_vote": round(held / votes / 64, 4),
        "refills_per_64_rays": round(64 * refills / rays, 3), "lanes_per_refill": round(refilled / max(refills, 1), 2),
        "box_pair_steps_per_ray": round(nlanes / rays, 3), "box_step_lane_util": round(nlanes / nsteps / 64, 4),
        "leaves_per_ray": round(llanes / rays, 3), "leaf_open_lane_util": round(llanes / max(lsteps, 1) / 64, 4),
        "triangles_per_ray": round(tlanes / rays, 3), "triangle_lane_util": round(tlanes / max(titer, 1) / 64, 4),
        "wave_steps_per_64_rays": {"box": round(64 * nsteps / rays, 2), "leaf": round(64 * lsteps / rays, 2), "tri": round(64 * titer / rays, 2)},
    }
print(json.dumps({"config": cfg, "scene": name, "frames": frames, "leaves": int(st.leaves_used), "extend_variant": int(st.extend_variant),
                  "shadow_variant": int(st.shadow_variant), "retraced": int(st.verify_failed), **res}, indent=1))
