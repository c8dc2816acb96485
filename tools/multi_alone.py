import json, os, sys, time
sys.path.insert(0, "wgpu-path-tracing_amd")
from ptmi import layout, native, scenes
W, H, F = 3840, 2160, 64
sc = scenes.make("cornell"); kw = dict(aperture=0.05, focus_distance=2.8)
which = sys.argv[1]
def run(o, f0):
    o.synchronize(); o.reset_stats()
    t = time.time(); o.dispatch(layout.make_camera(W, H, frame_index=f0, **kw), F); o.synchronize(); dt = time.time() - t
    return round(o.stats().segments / dt / 1e6, 1)
if which == "plain": o = native.Context(0)
elif which == "multi_rccl": o = native.MultiContext([0])
else: o = native.MultiContext([0], loopback=True)
o.upload_scene(sc); o.resize(W, H); o.set_options(max_bounces=8, do_mis=1, frames_per_batch=8)
run(o, 0)
print(which, [run(o, k * F) for k in range(1, 5)], (o.options().tile_strip, o.options().tile_parts) if which != "plain" else "")
