import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wgpu-path-tracing_amd")
for p in (PKG, os.path.dirname(os.path.abspath(__file__)), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # native libraries are built in-tree by __graft_entry__.build(); build lazily if a test run starts without them
    need = [os.path.join(PKG, "lib", "libptmi_scene.so"), os.path.join(ROOT, "oracle", "build", "libpt_oracle.so"),
            os.path.join(ROOT, "oracle", "build", "libpt_oracle_strict.so"), os.path.join(ROOT, "oracle", "build", "libpt_literal.so")]
    if not all(os.path.exists(p) for p in need):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", PKG, "scene"], stdout=subprocess.DEVNULL)


_scene_cache = {}


def get_scene(name):
    from ptmi import scenes
    if name not in _scene_cache:
        _scene_cache[name] = scenes.make(name)
    return _scene_cache[name]


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle(strict=False)


@pytest.fixture(scope="session")
def oracle_strict():
    from oracle_lib import Oracle
    return Oracle(strict=True)


@pytest.fixture(scope="session")
def oracle_literal():
    """oracle/pt_literal.c: the reference's shader restated in its own shape, sharing no code with pt_oracle.c"""
    from oracle_lib import Oracle
    return Oracle(literal=True)


@pytest.fixture(scope="session")
def scene_factory():
    return get_scene


@pytest.fixture(scope="session")
def gpu_ctx():
    """One HIP context for the whole GPU test session (fails loudly when there is no GPU)."""
    from ptmi import native
    ctx = native.Context(0)
    # PTMI_TEST_LEAVES=1 runs the whole GPU suite over the reference's leaves (ptmi_options.leaves; the default, 2, is the library's
    # own): tests change options through get -> modify -> set, so the mode stays unless a test names it
    if os.environ.get("PTMI_TEST_LEAVES"):
        ctx.set_options(leaves=int(os.environ["PTMI_TEST_LEAVES"]))
    yield ctx
    ctx.close()
