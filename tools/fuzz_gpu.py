#!/usr/bin/env python3
"""Heavier one-off version of tests/test_gpu_parity.py::test_random_scene_fuzz: more seeds, more pixels.
usage (GPU box): python tools/fuzz_gpu.py [n_seeds=20] [width=512] [height=384] [frames=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ptmi import layout, native, scenes
from oracle_lib import Oracle

n_seeds, W, H, frames = (int(a) for a in (sys.argv[1:] + ["20", "512", "384", "8"][len(sys.argv) - 1:])[:4])
oracle, ctx = Oracle(), native.Context(0)
bad = 0
for seed in range(n_seeds):
    sc = scenes.random_soup(seed, n_tris=400 + 150 * (seed % 5))
    cam = layout.make_camera(W, H, aperture=0.03 * (seed % 3), focus_distance=2.5, frame_index=seed * 7)
    t0 = time.time(); ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=seed % 4 != 3); t_cpu = time.time() - t0
    # (own leaves | the reference's leaves, rebuilt hierarchy | tree as uploaded) x (LDS | global quantised | global exact) x (one stream | shadow stream)
    for keep, leaves, trav, overlap in (
            (0, 2, native.TRAVERSAL_AUTO, 1), (0, 2, native.TRAVERSAL_GLOBAL, 0), (0, 2, native.TRAVERSAL_GLOBAL_EXACT, 1), (0, 2, native.TRAVERSAL_LDS, 1),
            (0, 1, native.TRAVERSAL_AUTO, 1), (1, 1, native.TRAVERSAL_GLOBAL, 0), (0, 1, native.TRAVERSAL_GLOBAL, 1), (0, 1, native.TRAVERSAL_GLOBAL_EXACT, 0)):
        ctx.set_options(keep_reference_tree=keep, leaves=leaves)
        ctx.upload_scene(sc); ctx.resize(W, H)
        ctx.set_options(max_bounces=8, do_mis=int(seed % 4 != 3), tile_y0=0, tile_y1=0, frames_per_batch=[0, 3][seed % 2], cull=1,
                        traversal=trav, overlap=overlap)
        ctx.reset_stats()
        try:
            ctx.dispatch(cam, frames)
        except native.PtmiError as e:             # a soup whose tree is too deep for the full LDS image: the library's own choice instead
            if "does not fit" not in str(e):
                raise
            ctx.set_options(traversal=native.TRAVERSAL_AUTO)
            ctx.dispatch(cam, frames)
        got = ctx.read_output(); st = ctx.stats()
        a, b = got.view(np.uint32), ref.view(np.uint32)
        diff = (a != b) & ~(np.isnan(got) & np.isnan(ref))
        ok = not diff.any() and (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays)
        bad += not ok
        print(f"seed {seed:3d} keep {keep} leaves {leaves} trav {trav} overlap {overlap} retraced {st.verify_failed} tris {len(sc.tris):5d} segs {st.segments:9d} nan {int(np.isnan(ref).sum()):6d} "
              f"cpu {t_cpu:5.2f}s  {'OK' if ok else 'MISMATCH %d floats, counts %s vs %s' % (diff.sum(), (st.segments, st.shadow_rays), (ost.segments, ost.shadow_rays))}", flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
