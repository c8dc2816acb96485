"""ctypes binding of libptmi.so (include/ptmi.h) — the HIP wavefront path tracer.

No fallback: if the library or a gfx950 device is missing, `Context()` raises.
"""
import ctypes
import os

import numpy as np

from . import layout

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
LIB_PATH = os.environ.get("PTMI_LIB") or os.path.join(_LIB_DIR, "libptmi.so")   # PTMI_LIB: A/B another build of the same ABI

TRAVERSAL_AUTO, TRAVERSAL_GLOBAL, TRAVERSAL_LDS, TRAVERSAL_GLOBAL_EXACT = 0, 1, 2, 3
ATLAS_RGBA16F, ATLAS_RGBA32F = 1, 2

# every symbol include/ptmi.h declares
EXPORTS = [
    "ptmi_abi_version", "ptmi_create", "ptmi_destroy", "ptmi_last_error", "ptmi_upload_scene",
    "ptmi_upload_atlas", "ptmi_resize", "ptmi_set_options", "ptmi_get_options", "ptmi_dispatch",
    "ptmi_synchronize", "ptmi_read_output", "ptmi_write_output", "ptmi_output_device_ptr",
    "ptmi_bind_output_device", "ptmi_set_stream", "ptmi_blit", "ptmi_get_stats", "ptmi_reset_stats",
    "ptmi_debug_raygen", "ptmi_debug_intersect", "ptmi_debug_occluded", "ptmi_debug_math", "ptmi_debug_exact_math", "ptmi_get_size",
    "ptmi_debug_image_stats", "ptmi_debug_build_image", "ptmi_throttle", "ptmi_multi_throttle",
    "ptmi_multi_create", "ptmi_multi_destroy", "ptmi_multi_last_error", "ptmi_multi_count", "ptmi_multi_context",
    "ptmi_multi_upload_scene", "ptmi_multi_upload_atlas", "ptmi_multi_resize", "ptmi_multi_set_options", "ptmi_multi_get_options",
    "ptmi_multi_dispatch", "ptmi_multi_gather", "ptmi_multi_synchronize", "ptmi_multi_read_output", "ptmi_multi_write_output",
    "ptmi_multi_blit", "ptmi_multi_get_stats", "ptmi_multi_reset_stats", "ptmi_multi_gather_ms",
]
MULTI_LOOPBACK = 1
ABI_VERSION = 4


class PtmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptmi error {code}: {msg}")
        self.code = code


class Options(ctypes.Structure):
    _fields_ = [("max_bounces", ctypes.c_uint32), ("do_mis", ctypes.c_uint32),
                ("tile_y0", ctypes.c_uint32), ("tile_y1", ctypes.c_uint32),
                ("frames_per_batch", ctypes.c_uint32), ("traversal", ctypes.c_uint32),
                ("cull", ctypes.c_uint32), ("timing", ctypes.c_uint32), ("keep_reference_tree", ctypes.c_uint32),
                ("tile_parts", ctypes.c_uint32), ("tile_part", ctypes.c_uint32), ("tile_strip", ctypes.c_uint32),
                ("perf_mode", ctypes.c_uint32), ("reserved_a", ctypes.c_uint32), ("overlap", ctypes.c_uint32),
                ("reserved_b", ctypes.c_uint32 * 4), ("tree_builder", ctypes.c_uint32),
                ("leaves", ctypes.c_uint32), ("leaf_tris", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 1)]


class Stats(ctypes.Structure):
    _fields_ = [("paths", ctypes.c_uint64), ("segments", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64),
                ("dispatches", ctypes.c_uint64), ("frames", ctypes.c_uint64),
                ("segments_by_bounce", ctypes.c_uint64 * 64),
                ("gpu_ms", ctypes.c_double), ("extend_ms", ctypes.c_double), ("extend_launches", ctypes.c_uint64),
                ("shade_ms", ctypes.c_double), ("shadow_ms", ctypes.c_double),
                ("bvh_depth", ctypes.c_uint32), ("traversal_used", ctypes.c_uint32),
                ("frames_per_batch_used", ctypes.c_uint32), ("radiance_stride_bytes", ctypes.c_uint32),
                ("shadow_traced", ctypes.c_uint64), ("shade_launches", ctypes.c_uint64), ("shadow_launches", ctypes.c_uint64),
                ("raygen_ms", ctypes.c_double), ("compact_ms", ctypes.c_double), ("accumulate_ms", ctypes.c_double),
                ("upload_ms", ctypes.c_double), ("upload_tree_ms", ctypes.c_double), ("upload_copy_ms", ctypes.c_double),
                ("leaves_used", ctypes.c_uint32), ("leaf_tris_used", ctypes.c_uint32),
                ("extend_variant", ctypes.c_uint32), ("shadow_variant", ctypes.c_uint32), ("verify_failed", ctypes.c_uint64),
                ("reserved_stats", ctypes.c_uint32 * 2)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("segments_by_bounce", "reserved_stats")}
        d["segments_by_bounce"] = [int(v) for v in self.segments_by_bounce if v]
        return d


_lib = None


def load():
    """Loads libptmi.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtmiError(-2, f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build`")
        L = ctypes.CDLL(LIB_PATH)
        L.ptmi_last_error.restype = ctypes.c_char_p
        L.ptmi_last_error.argtypes = [ctypes.c_void_p]
        L.ptmi_output_device_ptr.restype = ctypes.c_void_p
        L.ptmi_output_device_ptr.argtypes = [ctypes.c_void_p]
        L.ptmi_create.argtypes = [ctypes.c_int, ctypes.c_void_p]
        for name in EXPORTS:
            getattr(L, name)
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        L.ptmi_destroy.argtypes = [vp]
        L.ptmi_upload_scene.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32]
        L.ptmi_upload_atlas.argtypes = [vp, vp, u32, u32, ctypes.c_int]
        L.ptmi_resize.argtypes = [vp, u32, u32]
        L.ptmi_set_options.argtypes = [vp, vp]
        L.ptmi_get_options.argtypes = [vp, vp]
        L.ptmi_dispatch.argtypes = [vp, vp, u32]
        L.ptmi_synchronize.argtypes = [vp]
        L.ptmi_read_output.argtypes = [vp, vp, sz]
        L.ptmi_write_output.argtypes = [vp, vp, sz]
        L.ptmi_bind_output_device.argtypes = [vp, vp, sz]
        L.ptmi_set_stream.argtypes = [vp, vp]
        L.ptmi_blit.argtypes = [vp, vp, sz, vp, sz]
        L.ptmi_get_size.argtypes = [vp, vp, vp]
        L.ptmi_debug_image_stats.argtypes = [vp, u32, vp, u32, vp]
        L.ptmi_debug_build_image.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp, vp, vp]
        if L.ptmi_abi_version() != ABI_VERSION:
            raise PtmiError(-1, f"{LIB_PATH} has ABI {L.ptmi_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
        L.ptmi_get_stats.argtypes = [vp, vp]
        L.ptmi_reset_stats.argtypes = [vp]
        L.ptmi_debug_raygen.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp]
        L.ptmi_debug_intersect.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        L.ptmi_debug_occluded.argtypes = [vp, u32, vp, vp, vp, vp]
        L.ptmi_debug_math.argtypes = [vp, ctypes.c_int, u32, vp, vp, vp, vp]
        L.ptmi_debug_exact_math.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        L.ptmi_throttle.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_uint32)]
        L.ptmi_multi_throttle.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_uint32)]
        L.ptmi_multi_last_error.restype = ctypes.c_char_p
        L.ptmi_multi_last_error.argtypes = [vp]
        L.ptmi_multi_context.restype = vp
        L.ptmi_multi_context.argtypes = [vp, ctypes.c_int]
        L.ptmi_multi_create.argtypes = [ctypes.c_int, vp, u32, vp]
        L.ptmi_multi_count.argtypes = [vp]
        for name in ("destroy", "gather", "synchronize", "reset_stats"):
            getattr(L, "ptmi_multi_" + name).argtypes = [vp]
        L.ptmi_multi_upload_scene.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32]
        L.ptmi_multi_upload_atlas.argtypes = [vp, vp, u32, u32, ctypes.c_int]
        L.ptmi_multi_resize.argtypes = [vp, u32, u32]
        L.ptmi_multi_set_options.argtypes = [vp, vp]
        L.ptmi_multi_get_options.argtypes = [vp, vp]
        L.ptmi_multi_dispatch.argtypes = [vp, vp, u32]
        L.ptmi_multi_read_output.argtypes = [vp, vp, sz]
        L.ptmi_multi_write_output.argtypes = [vp, vp, sz]
        L.ptmi_multi_blit.argtypes = [vp, vp, sz, vp, sz]
        L.ptmi_multi_get_stats.argtypes = [vp, vp]
        L.ptmi_multi_gather_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def image_stats(scene):
    """Host-only report on the traversal image ptmi_upload_scene would build (include/ptmi.h: ptmi_debug_image_stats)."""
    L = load()
    out = (ctypes.c_double * 8)()
    rc = L.ptmi_debug_image_stats(_p(scene.tris), len(scene.tris), _p(scene.nodes), len(scene.nodes), out)
    if rc != 0:
        raise PtmiError(rc, L.ptmi_last_error(None).decode())
    keys = ("wide_nodes", "leaves", "depth", "quantised_nodes", "stream_dwords", "containment_violations",
            "mean_area_growth", "stream_mismatches")
    return dict(zip(keys, list(out)))


class ImageInfo(ctypes.Structure):
    _fields_ = [("leaves_used", ctypes.c_uint32), ("n_wnodes", ctypes.c_uint32), ("n_tris", ctypes.c_uint32),
                ("root_ref", ctypes.c_uint32), ("depth", ctypes.c_uint32), ("n_leaves", ctypes.c_uint32),
                ("max_leaf_tris", ctypes.c_uint32), ("quantised", ctypes.c_uint32),
                ("root_min", ctypes.c_float * 3), ("root_max", ctypes.c_float * 3),
                ("pad", ctypes.c_float), ("safe_origin", ctypes.c_float),
                ("q_origin", ctypes.c_float * 3), ("q_scale", ctypes.c_float * 3), ("ref_depth", ctypes.c_uint32)]


def build_image(scene, leaves=0, leaf_tris=0, keep_reference_tree=0):
    """Host-only: the traversal image ptmi_upload_scene would build (include/ptmi.h: ptmi_debug_build_image), as numpy arrays:
    (info, wnodes [n, 16] f32, qnodes [n, 8] u32 or None, tripos [m, 12] f32, leafbox [n_triangles, 8] f32 or None)."""
    L = load()
    o = Options()
    o.leaves, o.leaf_tris, o.keep_reference_tree = leaves, leaf_tris, keep_reference_tree
    info = ImageInfo()
    args = (_p(scene.tris), len(scene.tris), _p(scene.nodes), len(scene.nodes), ctypes.byref(o), ctypes.byref(info))
    rc = L.ptmi_debug_build_image(*args, None, None, None, None)
    if rc != 0:
        raise PtmiError(rc, L.ptmi_last_error(None).decode())
    wn = np.zeros((info.n_wnodes, 16), np.float32)
    qn = np.zeros((info.n_wnodes, 8), np.uint32) if info.quantised else None
    tp = np.zeros((info.n_tris, 12), np.float32)
    lb = np.zeros((len(scene.tris), 8), np.float32) if info.leaves_used == 2 else None
    rc = L.ptmi_debug_build_image(*args, _p(wn), _p(qn), _p(tp), _p(lb))
    if rc != 0:
        raise PtmiError(rc, L.ptmi_last_error(None).decode())
    return info, wn, qn, tp, lb


class Context:
    """One device context = the reference Renderer's GPU resources (bind group 0)."""

    def __init__(self, device=0):
        self.L = load()
        h = ctypes.c_void_p()
        rc = self.L.ptmi_create(device, ctypes.byref(h))
        if rc != 0:
            raise PtmiError(rc, self.L.ptmi_last_error(None).decode())
        self.h = h
        self.width = self.height = 0

    def _ck(self, rc):
        if rc != 0:
            raise PtmiError(rc, self.L.ptmi_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.ptmi_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- resources -----------------------------------------------------------
    def upload_scene(self, scene):
        for a, dt in ((scene.tris, layout.TRIANGLE), (scene.mats, layout.MATERIAL),
                      (scene.nodes, layout.BVH_NODE), (scene.lights, layout.LIGHT)):
            assert a.dtype == dt and a.flags.c_contiguous
        self._ck(self.L.ptmi_upload_scene(self.h, _p(scene.tris), len(scene.tris), _p(scene.mats), len(scene.mats),
                                          _p(scene.nodes), len(scene.nodes), _p(scene.lights), len(scene.lights)))
        a = scene.atlas
        if a is None:
            self._ck(self.L.ptmi_upload_atlas(self.h, None, 0, 0, 0))
        else:
            assert a.ndim == 3 and a.shape[2] == 4 and a.flags.c_contiguous
            fmt = ATLAS_RGBA16F if a.dtype == np.float16 else ATLAS_RGBA32F
            self._ck(self.L.ptmi_upload_atlas(self.h, _p(a), a.shape[1], a.shape[0], fmt))

    def resize(self, width, height):
        self._ck(self.L.ptmi_resize(self.h, width, height))
        self.width, self.height = width, height

    def set_options(self, **kw):
        o = Options()
        self._ck(self.L.ptmi_get_options(self.h, ctypes.byref(o)))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, int(v))
        self._ck(self.L.ptmi_set_options(self.h, ctypes.byref(o)))

    def options(self):
        o = Options()
        self._ck(self.L.ptmi_get_options(self.h, ctypes.byref(o)))
        return o

    # -- the compute pass ------------------------------------------------------
    def dispatch(self, camera, n_frames=1):
        assert camera.dtype == layout.CAMERA
        self._ck(self.L.ptmi_dispatch(self.h, _p(camera), n_frames))

    def synchronize(self):
        self._ck(self.L.ptmi_synchronize(self.h))

    def throttle(self, max_in_flight=0xFFFFFFFF):
        """Blocks until at most max_in_flight dispatches are unfinished (default: only polls); returns how many are."""
        n = ctypes.c_uint32(0)
        self._ck(self.L.ptmi_throttle(self.h, max_in_flight, ctypes.byref(n)))
        return int(n.value)

    def read_output(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._ck(self.L.ptmi_read_output(self.h, _p(out), out.size))
        return out

    def write_output(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        self._ck(self.L.ptmi_write_output(self.h, _p(arr), arr.size))

    def output_device_ptr(self):
        return self.L.ptmi_output_device_ptr(self.h)

    def bind_output_device(self, ptr, nbytes):
        self._ck(self.L.ptmi_bind_output_device(self.h, ctypes.c_void_p(ptr), nbytes))

    def set_stream(self, stream_handle):
        self._ck(self.L.ptmi_set_stream(self.h, ctypes.c_void_p(stream_handle)))

    def blit(self, want_f32=True, want_rgba8=True):
        """The reference's blit pass (blit.wgsl): (canvas float RGBA or None, canvas uint8 RGBA or None), row 0 = top."""
        f = np.empty((self.height, self.width, 4), np.float32) if want_f32 else None
        b = np.empty((self.height, self.width, 4), np.uint8) if want_rgba8 else None
        self._ck(self.L.ptmi_blit(self.h, _p(f), 0 if f is None else f.size, _p(b), 0 if b is None else b.size))
        return f, b

    def stats(self):
        s = Stats()
        self._ck(self.L.ptmi_get_stats(self.h, ctypes.byref(s)))
        return s

    def reset_stats(self):
        self._ck(self.L.ptmi_reset_stats(self.h))

    # -- per-stage entry points ------------------------------------------------
    def debug_raygen(self, camera, xs, ys, frames):
        xs, ys, frames = (np.ascontiguousarray(a, np.uint32) for a in (xs, ys, frames))
        n = len(xs)
        o, d, rng = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        self._ck(self.L.ptmi_debug_raygen(self.h, _p(camera), n, _p(xs), _p(ys), _p(frames), _p(o), _p(d), _p(rng)))
        return o, d, rng

    def debug_intersect(self, o, d):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        n = len(o)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        tri = np.zeros(n, np.uint32)
        self._ck(self.L.ptmi_debug_intersect(self.h, n, _p(o), _p(d), _p(t), _p(tri), _p(u), _p(v)))
        return t, tri, u, v

    def debug_occluded(self, o, d, dist):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        occ = np.zeros(len(o), np.uint8)
        self._ck(self.L.ptmi_debug_occluded(self.h, len(o), _p(o), _p(d), _p(dist), _p(occ)))
        return occ

    def debug_math(self, op, a, b=None, c=None):
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        c = None if c is None else np.ascontiguousarray(c, np.float32)
        out = np.zeros_like(a)
        self._ck(self.L.ptmi_debug_math(self.h, op, a.size, _p(a), _p(b), _p(c), _p(out)))
        return out

    def debug_exact_math(self, which):
        """(inputs among all 2^32 float patterns whose short-form result differs from the IEEE expansion, smallest such pattern);
        which = 0: 1/x, 1: sqrt(x), 2: the triangle test's 1/x (|x| >= 1e-6). include/ptmi.h."""
        n, first = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self._ck(self.L.ptmi_debug_exact_math(self.h, which, ctypes.byref(n), ctypes.byref(first)))
        return int(n.value), int(first.value)


class MultiContext:
    """Several devices of one node behind one handle (include/ptmi.h ptmi_multi_*): the frame's rows are dealt out as
    interleaved strips, every device accumulates its own, gather() assembles the frame on the first device through RCCL.
    loopback=True replaces the collective with device-to-device copies, so that one device can stand in for several."""

    def __init__(self, devices, loopback=False):
        self.L = load()
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.L.ptmi_multi_create(len(devices), devs, MULTI_LOOPBACK if loopback else 0, ctypes.byref(h))
        if rc != 0:
            raise PtmiError(rc, self.L.ptmi_multi_last_error(None).decode())
        self.h = h
        self.n = len(devices)
        self.width = self.height = 0

    def _ck(self, rc):
        if rc != 0:
            raise PtmiError(rc, self.L.ptmi_multi_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.ptmi_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_scene(self, scene):
        self._ck(self.L.ptmi_multi_upload_scene(self.h, _p(scene.tris), len(scene.tris), _p(scene.mats), len(scene.mats),
                                                _p(scene.nodes), len(scene.nodes), _p(scene.lights), len(scene.lights)))
        a = scene.atlas
        if a is None:
            self._ck(self.L.ptmi_multi_upload_atlas(self.h, None, 0, 0, 0))
        else:
            fmt = ATLAS_RGBA16F if a.dtype == np.float16 else ATLAS_RGBA32F
            self._ck(self.L.ptmi_multi_upload_atlas(self.h, _p(a), a.shape[1], a.shape[0], fmt))

    def resize(self, width, height):
        self._ck(self.L.ptmi_multi_resize(self.h, width, height))
        self.width, self.height = width, height

    def set_options(self, **kw):
        o = Options()
        self._ck(self.L.ptmi_multi_get_options(self.h, ctypes.byref(o)))
        o.tile_parts = o.tile_part = 0
        if "tile_strip" not in kw:
            o.tile_strip = 0
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, int(v))
        self._ck(self.L.ptmi_multi_set_options(self.h, ctypes.byref(o)))

    def options(self):
        o = Options()
        self._ck(self.L.ptmi_multi_get_options(self.h, ctypes.byref(o)))
        return o

    def dispatch(self, camera, n_frames=1):
        assert camera.dtype == layout.CAMERA
        self._ck(self.L.ptmi_multi_dispatch(self.h, _p(camera), n_frames))

    def gather(self):
        self._ck(self.L.ptmi_multi_gather(self.h))

    def synchronize(self):
        self._ck(self.L.ptmi_multi_synchronize(self.h))

    def read_output(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._ck(self.L.ptmi_multi_read_output(self.h, _p(out), out.size))
        return out

    def write_output(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        self._ck(self.L.ptmi_multi_write_output(self.h, _p(arr), arr.size))

    def blit(self):
        b = np.empty((self.height, self.width, 4), np.uint8)
        self._ck(self.L.ptmi_multi_blit(self.h, None, 0, _p(b), b.size))
        return b

    def stats(self):
        s = Stats()
        self._ck(self.L.ptmi_multi_get_stats(self.h, ctypes.byref(s)))
        return s

    def reset_stats(self):
        self._ck(self.L.ptmi_multi_reset_stats(self.h))

    def gather_ms(self):
        ms = ctypes.c_double(-1.0)
        self._ck(self.L.ptmi_multi_gather_ms(self.h, ctypes.byref(ms)))
        return ms.value
