"""The C-ABI shared library: loads, exports every symbol include/ptmi.h declares, and — on a
machine without a GPU — refuses to create a context (no CPU fallback behind the product ABI)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wgpu-path-tracing_amd")
LIB = os.path.join(PKG, "lib", "libptmi.so")


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ptmi_[a-z_0-9]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", PKG, "all"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(lib):
    names = declared("ptmi.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptmi.h but not exported by libptmi.so"
    from ptmi import native
    assert sorted(native.EXPORTS) == names


def test_scene_header_symbols_exported():
    from ptmi import scene_host
    L = scene_host.lib()
    for n in declared("ptmi_scene.h"):
        assert hasattr(L, n)


def test_abi_version_and_struct_sizes(lib):
    from ptmi import native
    assert lib.ptmi_abi_version() == native.ABI_VERSION == 4
    assert ctypes.sizeof(native.Options) == 23 * 4      # ABI 4: the size of ABI 3 (five of its options are reserved words now; leaves, leaf_tris are new)
    # ptmi_stats: 5 + 64 u64, 2 f64 + u64 + 2 f64, 4 u32; ABI 2 adds 3 u64 + 6 f64
    assert ctypes.sizeof(native.Stats) == (5 + 64) * 8 + 5 * 8 + 16 + 9 * 8 + 16 + 8 + 8      # ABI 4: leaves_used, leaf_tris_used, two variants, verify_failed, two reserved words


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / offsetof of every struct the binding mirrors, as a C compiler lays out include/ptmi.h."""
    from ptmi import native
    src = tmp_path / "sizes.c"
    fields = {"ptmi_options": ["max_bounces", "timing", "tile_strip", "perf_mode", "reserved_a", "overlap", "reserved_b", "tree_builder", "leaves", "leaf_tris", "reserved"],
              "ptmi_stats": ["paths", "segments_by_bounce", "gpu_ms", "extend_launches", "bvh_depth", "shadow_traced",
                             "raygen_ms", "upload_copy_ms", "leaves_used", "leaf_tris_used", "extend_variant", "shadow_variant", "verify_failed"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ptmi.h"', 'int main(void) {']
    for st, fs in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        lines += [f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fs]
    lines += ['return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("ptmi_options", native.Options), ("ptmi_stats", native.Stats)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f in fields[st]:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, f"{st}.{f}"


def test_no_gpu_means_no_context(lib):
    """Without a device the product path must fail loudly, not fall back."""
    lib.ptmi_last_error.restype = ctypes.c_char_p
    ctx = ctypes.c_void_p()
    rc = lib.ptmi_create(0, ctypes.byref(ctx))
    if rc == 0:                                     # running on a GPU box: fine, clean up
        lib.ptmi_destroy(ctx)
        pytest.skip("a GPU is present")
    assert rc == -2 and not ctx.value
    assert b"no CPU backend" in lib.ptmi_last_error(None)
    from ptmi import native
    with pytest.raises(native.PtmiError):
        native.Context(0)


def test_product_does_not_reference_the_oracle():
    """Nothing under the product package may import, link or name the oracle."""
    bad = []
    for dp, _, files in os.walk(PKG):
        if os.sep + "lib" in dp or "node_modules" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".js", ".ts", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"pt_oracle|oracle_lib|oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "oracle" not in out
