"""ctypes binding of the CPU oracle (oracle/pt_oracle.h) — test infrastructure.

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class PtoScene(ctypes.Structure):
    _fields_ = [("tris", ctypes.c_void_p), ("n_tris", ctypes.c_uint32),
                ("mats", ctypes.c_void_p), ("n_mats", ctypes.c_uint32),
                ("nodes", ctypes.c_void_p), ("n_nodes", ctypes.c_uint32),
                ("lights", ctypes.c_void_p), ("n_lights", ctypes.c_uint32),
                ("atlas", ctypes.c_void_p), ("atlas_w", ctypes.c_uint32), ("atlas_h", ctypes.c_uint32),
                ("atlas_fmt", ctypes.c_int32)]


class PtoOptions(ctypes.Structure):
    _fields_ = [("max_bounces", ctypes.c_uint32), ("do_mis", ctypes.c_uint32),
                ("y0", ctypes.c_uint32), ("y1", ctypes.c_uint32), ("threads", ctypes.c_uint32)]


class PtoStats(ctypes.Structure):
    _fields_ = [("paths", ctypes.c_uint64), ("segments", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64),
                ("nodes_visited", ctypes.c_uint64), ("tris_tested", ctypes.c_uint64),
                ("closest_hits", ctypes.c_uint64), ("max_stack", ctypes.c_uint32), ("threads", ctypes.c_uint32),
                ("seconds", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, strict=False, literal=False):
        """strict: pt_oracle.c built with literal arithmetic; literal: oracle/pt_literal.c, the independent transcription in the
        reference's own shape (it offers the subset of entry points the literal comparisons use)"""
        name = "libpt_literal.so" if literal else "libpt_oracle_strict.so" if strict else "libpt_oracle.so"
        path = os.path.join(os.environ.get("PT_ORACLE_BUILD_DIR") or os.path.join(ORACLE_DIR, "build"), name)   # (a sanitizer build, tools/sanitize/run_oracle.sh)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(path)
        L.pto_seed.restype = ctypes.c_uint32
        L.pto_seed.argtypes = [ctypes.c_uint32] * 3
        L.pto_rand_int.restype = ctypes.c_uint32
        L.pto_rand_int.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        L.pto_sincos.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        L.pto_distribution_ggx.restype = ctypes.c_float
        L.pto_distribution_ggx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float]
        L.pto_power_heuristic.restype = ctypes.c_float
        L.pto_power_heuristic.argtypes = [ctypes.c_float] * 4
        L.pto_eval_bsdf.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.L = L
        self.strict = bool(L.pto_is_strict())
        self.literal = L.pto_is_strict() == 2
        assert self.literal == literal and self.strict == (strict or literal)

    # -- scene marshalling ---------------------------------------------------
    @staticmethod
    def scene_struct(scene):
        s = PtoScene()
        s.tris, s.n_tris = _ptr(scene.tris), len(scene.tris)
        s.mats, s.n_mats = _ptr(scene.mats), len(scene.mats)
        s.nodes, s.n_nodes = _ptr(scene.nodes), len(scene.nodes)
        s.lights, s.n_lights = _ptr(scene.lights), len(scene.lights)
        if scene.atlas is not None:
            a = scene.atlas
            assert a.dtype in (np.float16, np.float32) and a.ndim == 3 and a.shape[2] == 4 and a.flags.c_contiguous
            s.atlas, s.atlas_h, s.atlas_w = _ptr(a), a.shape[0], a.shape[1]
            s.atlas_fmt = 1 if a.dtype == np.float16 else 2
        return s

    # -- RNG -----------------------------------------------------------------
    def seed(self, x, y, frame):
        return self.L.pto_seed(x, y, frame)

    def rand(self, state, n):
        st = ctypes.c_uint32(state)
        states, words, vals = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
        self.L.pto_rand(ctypes.byref(st), n, _ptr(states), _ptr(words), _ptr(vals))
        return states, words, vals

    def rand_int(self, state, lo, hi):
        st = ctypes.c_uint32(state)
        k = self.L.pto_rand_int(ctypes.byref(st), lo, hi)
        return k, st.value

    def sincos(self, x):
        s, c = ctypes.c_float(), ctypes.c_float()
        self.L.pto_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
        return s.value, c.value

    # -- stages --------------------------------------------------------------
    def raygen(self, cam, xs, ys, frames):
        xs, ys, frames = (np.ascontiguousarray(a, np.uint32) for a in (xs, ys, frames))
        n = len(xs)
        o, d, rng = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        self.L.pto_raygen(_ptr(cam), n, _ptr(xs), _ptr(ys), _ptr(frames), _ptr(o), _ptr(d), _ptr(rng))
        return o, d, rng

    def intersect(self, scene, o, d):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        n = len(o)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        tri = np.zeros(n, np.uint32)
        st = PtoStats()
        s = self.scene_struct(scene)
        self.L.pto_intersect(ctypes.byref(s), n, _ptr(o), _ptr(d), _ptr(t), _ptr(tri), _ptr(u), _ptr(v), ctypes.byref(st))
        return t, tri, u, v, st

    def occluded(self, scene, o, d, dist):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        occ = np.zeros(len(o), np.uint8)
        s = self.scene_struct(scene)
        self.L.pto_occluded(ctypes.byref(s), len(o), _ptr(o), _ptr(d), _ptr(dist), _ptr(occ), None)
        return occ

    def render(self, scene, cam, n_frames, max_bounces=8, do_mis=1, out=None, y0=0, y1=0, threads=0):
        W, H = int(cam["width"]), int(cam["height"])
        if out is None:
            out = np.zeros((H, W, 4), np.float32)
        assert out.dtype == np.float32 and out.shape == (H, W, 4) and out.flags.c_contiguous
        opt = PtoOptions(max_bounces, do_mis, y0, y1, threads)
        st = PtoStats()
        s = self.scene_struct(scene)
        rc = self.L.pto_render(ctypes.byref(s), _ptr(cam), n_frames, ctypes.byref(opt), _ptr(out), ctypes.byref(st))
        assert rc == 0
        return out, st

    def blit(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        H, W = rgba.shape[:2]
        out = np.zeros_like(rgba)
        self.L.pto_blit(_ptr(rgba), W, H, _ptr(out))
        return out

    def trace_path(self, scene, cam, x, y, frame, max_bounces=8, do_mis=1):
        opt = PtoOptions(max_bounces, do_mis, 0, 0, 1)
        rad = np.zeros(3, np.float32)
        log = np.zeros((max_bounces + 1, 16), np.float32)
        s = self.scene_struct(scene)
        n = self.L.pto_trace_path(ctypes.byref(s), _ptr(cam), x, y, frame, ctypes.byref(opt), _ptr(rad), _ptr(log))
        return rad, log[:n]

    # -- probes ----------------------------------------------------------------
    def eval_bsdf(self, albedo, rough, metal, trans, ior, n, v, l, front=True):
        a, n, v, l = (np.ascontiguousarray(q, np.float32) for q in (albedo, n, v, l))
        out = np.zeros(4, np.float32)
        self.L.pto_eval_bsdf(_ptr(a), rough, metal, trans, ior, _ptr(n), _ptr(v), _ptr(l), int(front), _ptr(out))
        return out

    def distribution_ggx(self, n, h, rough):
        n, h = np.ascontiguousarray(n, np.float32), np.ascontiguousarray(h, np.float32)
        return self.L.pto_distribution_ggx(_ptr(n), _ptr(h), rough)

    def power_heuristic(self, nf, fp, ng, gp):
        return self.L.pto_power_heuristic(nf, fp, ng, gp)

    def cosine_direction(self, state):
        st = ctypes.c_uint32(state)
        out = np.zeros(3, np.float32)
        self.L.pto_cosine_direction(ctypes.byref(st), _ptr(out))
        return out, st.value

    def sample_ggx_normal(self, state, n, rough):
        st = ctypes.c_uint32(state)
        n = np.ascontiguousarray(n, np.float32)
        out = np.zeros(3, np.float32)
        self.L.pto_sample_ggx_normal.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
        self.L.pto_sample_ggx_normal(ctypes.byref(st), _ptr(n), rough, _ptr(out))
        return out, st.value
