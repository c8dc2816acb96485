#!/usr/bin/env python3
"""Times ptmi_upload_scene (validation + traversal-image build + copies) per scene on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
from ptmi import native, scenes
ctx = native.Context(0)
for name in sys.argv[1:] or ["cornell", "cornell_spheres", "grid_1m"]:
    t = time.time(); sc = scenes.make(name); t_make = time.time() - t
    for leaves, keep, tb in ((2, 0, 1), (1, 0, 1), (1, 0, 2), (1, 1, 1)):
        ctx.set_options(leaves=leaves, keep_reference_tree=keep, tree_builder=tb)
        t = time.time(); ctx.upload_scene(sc); dt = time.time() - t
        st = ctx.stats()
        print(f"{name:16s} triangles {len(sc.tris):8d} host prep {t_make:6.2f} s  upload(leaves={leaves}, keep_reference_tree={keep}, tree_builder={tb}) {dt:6.3f} s"
              f"  [library: total {st.upload_ms:8.1f} ms, rebuilt hierarchy {st.upload_tree_ms:8.1f} ms, copies {st.upload_copy_ms:7.1f} ms]", flush=True)
