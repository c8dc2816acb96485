"""ctypes binding of libptmi_scene.so (include/ptmi_scene.h): host-side scene
preparation — the partial quicksort, SAH-BVH builder and emissive-light list of
the reference (src/utils/arr.ts, src/renderer/bvh.ts, src/renderer/gpu.ts:121-138).
"""
import ctypes
import os

import numpy as np

from . import layout

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
_lib = None


class SceneError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_LIB_DIR, "libptmi_scene.so")
        if not os.path.exists(path):
            raise SceneError(f"{path} is missing: run `python __graft_entry__.py build` (or make -C wgpu-path-tracing_amd)")
        L = ctypes.CDLL(path)
        L.ptmi_scene_last_error.restype = ctypes.c_char_p
        L.ptmi_scene_bvh_node_bound.restype = ctypes.c_uint32
        L.ptmi_scene_bvh_node_bound.argtypes = [ctypes.c_uint32]
        L.ptmi_scene_sort_partially_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_int64, ctypes.c_int]
        L.ptmi_scene_build_bvh.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        L.ptmi_scene_set_threads.argtypes = [ctypes.c_int]
        L.ptmi_scene_set_threads.restype = None
        L.ptmi_scene_emissive_lights.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                                 ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SceneError(lib().ptmi_scene_last_error().decode())


def sort_partially(arr, start, end, descending=False):
    """arr.ts sortArrayPartially on a float64 array with (a,b)=>a-b (or b-a)."""
    a = np.ascontiguousarray(arr, dtype=np.float64)
    _check(lib().ptmi_scene_sort_partially_f64(a.ctypes.data, a.size, start, end, int(descending)))
    return a


def build_bvh(tris, max_leaf=4, bins=12, threads=0):
    """Sorts `tris` (TRIANGLE array) in place and returns (nodes, max_depth). threads: 1 = the reference's single
    loop, 0 = one per hardware thread; the result is byte-identical either way."""
    assert tris.dtype == layout.TRIANGLE and tris.flags.c_contiguous
    L = lib()
    L.ptmi_scene_set_threads(int(threads))
    cap = L.ptmi_scene_bvh_node_bound(len(tris))
    nodes = np.zeros(cap, layout.BVH_NODE)
    n = ctypes.c_uint32(0)
    depth = ctypes.c_uint32(0)
    _check(L.ptmi_scene_build_bvh(tris.ctypes.data, len(tris), max_leaf, bins, nodes.ctypes.data, cap,
                                  ctypes.byref(n), ctypes.byref(depth)))
    return nodes[: n.value].copy(), depth.value


def emissive_lights(tris, mats, punctual=None):
    """Light list in the reference's order: punctual lights, then one emissive
    light per emissive triangle (post-sort index)."""
    n0 = 0 if punctual is None else len(punctual)
    out = np.zeros(n0 + len(tris), layout.LIGHT)
    if n0:
        out[:n0] = punctual
    n = ctypes.c_uint32(n0)
    _check(lib().ptmi_scene_emissive_lights(tris.ctypes.data, len(tris), mats.ctypes.data, len(mats),
                                            out.ctypes.data, len(out), ctypes.byref(n)))
    return out[: n.value].copy()
