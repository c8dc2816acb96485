#!/usr/bin/env python3
"""CPU gate and CPU parity count for the library's own leaves (ptmi_options.leaves = 2). No GPU.

Rays: every ray of a real render of a BASELINE-config scene — camera rays, bounce rays, shadow rays, with the reference traversal's
result — recorded by the oracle (oracle/pt_oracle.c pto_render_tap; 8 bounces, MIS on). They are replayed by tools/own_sim.c, which
walks the image ptmi_debug_build_image returns with the kernels' own arithmetic (traverse_own.hip), once per variant:
  ref       the hierarchy rebuilt over the reference's leaves (leaves = 1, traverse.hip) — the baseline
  own K     own leaves of at most K triangles, exact nodes / quantised nodes
each with leaves tested at once ("imm": the best case of the kernels' majority scheduling) and after the whole descent ("def": the worst).
Printed per variant: box-pair steps, leaves and triangle tests per ray, the instruction estimate 65 (54 with the fused slab test, 66
quantised) x steps + 54 x triangles + 20 x leaves, its ratio to the baseline, how many results DIFFER from the reference traversal's,
and how many rays went the slow way (from the start / after a failed verification of the winner).

usage: tools/own_leaf_gate.py [scene ...] [--rays N] [--width W --height H --frames F] [--leaf-tris 1,2,4] [--json out.json]
       scenes: cornell cornell_spheres grid_1m feature_box (default: cornell cornell_spheres)"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ptmi import layout, native, scenes          # noqa: E402
from oracle_lib import Oracle, PtoOptions        # noqa: E402


class SimScene(ctypes.Structure):
    _fields_ = [("wn", ctypes.c_void_p), ("n_wn", ctypes.c_uint32), ("qn", ctypes.c_void_p),
                ("qo", ctypes.c_float * 3), ("qs", ctypes.c_float * 3),
                ("tp", ctypes.c_void_p), ("n_tp", ctypes.c_uint32), ("leafbox", ctypes.c_void_p),
                ("root_ref", ctypes.c_uint32), ("root_min", ctypes.c_float * 3), ("root_max", ctypes.c_float * 3),
                ("safe_origin", ctypes.c_float), ("tri_safe_dsum", ctypes.c_float),
                ("nodes", ctypes.c_void_p), ("n_nodes", ctypes.c_uint32), ("tris", ctypes.c_void_p), ("n_tris", ctypes.c_uint32)]


def sim_lib():
    src = os.path.join(ROOT, "tools", "own_sim.c")
    out = os.path.join(ROOT, "tools", "build", "libown_sim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                               "-I" + os.path.join(ROOT, "include"), "-o", out, src, "-lm"])
    L = ctypes.CDLL(out)
    L.own_sim_run.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                              ctypes.c_void_p]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Image:
    """A traversal image on the host + the struct own_sim.c reads (keeps the arrays alive)."""

    def __init__(self, scene, leaves, leaf_tris=0):
        self.scene = scene
        self.info, self.wn, self.qn, self.tp, self.lb = native.build_image(scene, leaves=leaves, leaf_tris=leaf_tris)
        e = self.tp[:, [4, 5, 6, 8, 9, 10]].astype(np.float64).reshape(-1, 3)
        emax2 = float((e * e).sum(axis=1).max()) if len(e) else 0.0
        s = SimScene()
        s.wn, s.n_wn, s.qn = _p(self.wn), self.info.n_wnodes, _p(self.qn)
        for k in range(3):
            s.qo[k], s.qs[k] = self.info.q_origin[k], self.info.q_scale[k]
            s.root_min[k], s.root_max[k] = self.info.root_min[k], self.info.root_max[k]
        s.tp, s.n_tp, s.leafbox = _p(self.tp), self.info.n_tris, _p(self.lb)
        s.root_ref = self.info.root_ref
        s.safe_origin = self.info.safe_origin if leaves == 2 else 3.0e38
        s.tri_safe_dsum = min(2.0 ** 98 / emax2, 3.0e38) if emax2 > 0 else 3.0e38
        s.nodes, s.n_nodes, s.tris, s.n_tris = _p(scene.nodes), len(scene.nodes), _p(scene.tris), len(scene.tris)
        self.s = s
        self.own = 1 if self.info.leaves_used == 2 else 0


def tap_rays(oracle, scene, cam, frames, y0, y1, max_rays, max_bounces=8, do_mis=1):
    """Every ray of the render of rows [y0, y1) x frames, with the reference traversal's result: [n, 9] float32."""
    L = oracle.L
    L.pto_render_tap.restype = ctypes.c_uint64
    L.pto_render_tap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    rec = np.zeros((max_rays, 9), np.float32)
    opt = PtoOptions(max_bounces, do_mis, y0, y1, 0)
    s = oracle.scene_struct(scene)
    n = L.pto_render_tap(ctypes.byref(s), _p(cam), frames, ctypes.byref(opt), _p(rec), max_rays)
    return rec[:min(n, max_rays)], n


def run(L, img, rec, quant, cull=1, deferred=0, want_diff=0):
    sums = np.zeros(12, np.uint64)
    diff = np.zeros(max(want_diff, 1), np.uint64)
    nd = ctypes.c_uint64(0)
    L.own_sim_run(ctypes.byref(img.s), len(rec), _p(rec), img.own, quant, cull, deferred, None, None, None, _p(sums),
                  _p(diff) if want_diff else None, want_diff, ctypes.byref(nd))
    return sums, diff[:nd.value]


def per_ray(sums, box_cost):
    nc, ns = max(int(sums[6]), 1), max(int(sums[7]), 1)
    c = [int(sums[0]) / nc, int(sums[1]) / nc, int(sums[2]) / nc]
    a = [int(sums[3]) / ns, int(sums[4]) / ns, int(sums[5]) / ns]
    est = lambda v: box_cost * v[0] + 54.0 * v[2] + 20.0 * v[1]
    return c, a, est(c), est(a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["cornell", "cornell_spheres"])
    ap.add_argument("--rays", type=float, default=2e6, help="rays per scene (whole rows of the render are taken until there are that many)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--leaf-tris", default="1,2,4")
    ap.add_argument("--json")
    ap.add_argument("--quick", action="store_true", help="own leaves of the default size only, both node formats, leaves at once")
    a = ap.parse_args()
    L = sim_lib()
    oracle = Oracle()
    report = {}
    for name in a.scenes:
        scene = scenes.grid_1m() if name == "grid_1m" else scenes.make(name)
        cam = layout.make_camera(a.width, a.height)
        t0 = time.time()
        variants = [("ref", 1, 0, 0)]
        for k in ([0] if a.quick else [int(x) for x in a.leaf_tris.split(",")]):
            variants += [(f"own{k or ''}", 2, k, 0), (f"own{k or ''}q", 2, k, 1)]
        runs = []                                   # (tag, image, quant, deferred, box cost, sums, differing rays)
        for tag, leaves, k, quant in variants:
            t1 = time.time()
            img = Image(scene, leaves, k)
            if quant and img.qn is None:
                print(f"  {tag}: no quantised image for this scene"); continue
            img.build_ms = (time.time() - t1) * 1e3
            for deferred in ((0,) if a.quick else (0, 1)):
                runs.append([tag, img, quant, deferred, 65.0 if leaves == 1 else (66.0 if quant else 54.0), np.zeros(12, np.uint64), []])
        # rows spread over the picture, a chunk of rays at a time, until the ray budget is met
        want, got, rows_done, n_shadow = int(a.rays), 0, 0, 0
        step = max(1, a.height // 64)
        rows = list(range(step // 2, a.height, step)) + [y for y in range(a.height) if (y - step // 2) % step]
        frame0 = 0
        while got < want:
            recs, chunk = [], 0
            while chunk < (1 << 23) and got + chunk < want:
                if rows_done == len(rows):          # the whole picture is used up: go on with the next frames
                    rows_done = 0; frame0 += a.frames
                cam["frame_index"] = frame0
                rec, n = tap_rays(oracle, scene, cam, a.frames, rows[rows_done], rows[rows_done] + 1, 1 << 22)
                recs.append(rec); chunk += len(rec); rows_done += 1
            rec = np.ascontiguousarray(np.concatenate(recs))
            got += len(rec); n_shadow += int((rec[:, 6] != 0).sum())
            for r in runs:
                sums, diff = run(L, r[1], rec, r[2], 1, r[3], want_diff=16)
                r[5] += sums
                r[6] += [rec[int(i)].copy() for i in diff[:4]]
            print(f"   ... {got} rays", flush=True)
        print(f"== {name}: {len(scene.tris)} triangles, {got} rays of a {a.width}x{a.height} render, frames {a.frames} at a time "
              f"({got - n_shadow} closest-hit, {n_shadow} shadow), {time.time() - t0:.1f} s", flush=True)
        out, base = {}, {}
        for tag, img, quant, deferred, box, sums, diffs in runs:
            c, s_, ec, es = per_ray(sums, box)
            if tag == "ref":
                base[deferred] = (ec, es)
            rc = ec / base[deferred][0] if deferred in base else float("nan")
            rs = es / base[deferred][1] if deferred in base and base[deferred][1] else float("nan")
            print(f"  {tag:6s} {'def' if deferred else 'imm'}  nodes {img.info.n_wnodes:7d} depth {img.info.depth:2d} | closest: steps {c[0]:6.2f} leaves {c[1]:5.2f} "
                  f"tris {c[2]:6.2f} est {ec:7.0f} ({rc:5.2f}) | shadow: steps {s_[0]:6.2f} leaves {s_[1]:5.2f} tris {s_[2]:6.2f} est {es:7.0f} ({rs:5.2f}) | "
                  f"differ {int(sums[8])} + {int(sums[9])}, slow {int(sums[10])}, retraced {int(sums[11])} | build {img.build_ms:.0f} ms", flush=True)
            out[f"{tag}_{'def' if deferred else 'imm'}"] = {
                "nodes": img.info.n_wnodes, "depth": img.info.depth, "closest": c, "shadow": s_, "est_closest": ec, "est_shadow": es,
                "ratio_closest": rc, "ratio_shadow": rs, "differ_closest": int(sums[8]), "differ_shadow": int(sums[9]),
                "slow": int(sums[10]), "retraced": int(sums[11]), "rays": got, "closest_rays": int(sums[6]), "shadow_rays": int(sums[7])}
            for r in diffs[:4]:
                print(f"      differing ray: o {r[0:3]} d {r[3:6]} dist {r[6]} reference t {r[7]} tri {r[8:9].view(np.uint32)[0]}")
        report[name] = out
    if a.json:
        with open(a.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
