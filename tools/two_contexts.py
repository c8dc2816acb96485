#!/usr/bin/env python3
"""tools/two_contexts.py [mode]: why is the SECOND context of a process 5 % slower? Msamples/s of 64-frame dispatches of BASELINE configs[4]'s
frame. mode a (default): two contexts alive, alternating; b: a first context created and destroyed, then a second one measured alone;
c: a first context alive but never used, second measured; d: like a, with 8 dummy streams created before the second context."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
from ptmi import layout, native, scenes  # noqa: E402
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
W, H, F = 3840, 2160, 64
sc = scenes.make("cornell"); kw = dict(aperture=0.05, focus_distance=2.8)


def make():
    o = native.Context(0); o.upload_scene(sc); o.resize(W, H); o.set_options(max_bounces=8, do_mis=1, frames_per_batch=8); return o


def run(o, f0):
    o.synchronize(); o.reset_stats()
    t = time.time(); o.dispatch(layout.make_camera(W, H, frame_index=f0, **kw), F); o.synchronize(); dt = time.time() - t
    return round(o.stats().segments / dt / 1e6, 1)


out = {"mode": mode}
if mode == "a":
    a, b = make(), make(); run(a, 0); run(b, 0)
    out["first"] = [run(a, k * F) for k in range(1, 4)]; out["second"] = [run(b, k * F) for k in range(1, 4)]
elif mode == "b":
    a = make(); run(a, 0); out["first"] = [run(a, k * F) for k in range(1, 4)]; a.close()
    b = make(); run(b, 0); out["second_after_first_destroyed"] = [run(b, k * F) for k in range(1, 4)]
elif mode == "c":
    a = native.Context(0)
    b = make(); run(b, 0); out["second_first_idle"] = [run(b, k * F) for k in range(1, 4)]
elif mode == "d":
    import torch
    a = make(); run(a, 0)
    dummies = [torch.cuda.Stream() for _ in range(3)]
    b = make(); run(b, 0)
    out["first"] = [run(a, k * F) for k in range(1, 4)]; out["second_after_3_dummy_streams"] = [run(b, k * F) for k in range(1, 4)]
print(json.dumps(out))
