"""ptmi_options.perf_mode = 1: `shade` built with the device's fast reciprocal / square root, FMA contraction and flushed
denormals (the same shade.hip, Makefile FAST_SHADE_FLAGS). Not the headline and not bit-exact by design; the gate is
the one SURVEY.md §8(c) proposes for a perf mode — same RNG streams, per 16x16 tile within 3 standard errors of the
oracle's Monte-Carlo mean — applied against BOTH oracle builds (the contract build the parity mode equals bit for bit,
and the literal transcription of src/shader/pt.wgsl:638-762), plus image means and counters."""
import numpy as np
import pytest

from ptmi import layout, scenes
from test_gpu_strict import compare, literal_statistics

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cornell", "feature_box"])
def test_perf_mode_is_statistically_the_same_image(gpu_ctx, oracle, oracle_literal, name):
    sc = scenes.make(name)
    W = H = 256
    frames = 64
    cam = layout.make_camera(W, H)
    gpu_ctx.upload_scene(sc)
    out = {}
    for mode in (0, 1):
        gpu_ctx.resize(W, H)
        gpu_ctx.set_options(max_bounces=8, do_mis=1, tile_y0=0, tile_y1=0, tile_parts=0, frames_per_batch=0, perf_mode=mode)
        gpu_ctx.reset_stats()
        gpu_ctx.dispatch(cam, frames)
        out[mode] = (gpu_ctx.read_output(), gpu_ctx.stats())
    gpu_ctx.set_options(perf_mode=0)
    exact, st0 = out[0]
    fast, st1 = out[1]
    ref, ost = oracle.render(sc, cam, frames)
    assert np.array_equal(exact.view(np.uint32), ref.view(np.uint32))           # parity mode is untouched by the second build
    assert st1.paths == st0.paths and abs(st1.segments / st0.segments - 1) < 1e-3 and abs(st1.shadow_rays / st0.shadow_rays - 1) < 1e-3
    assert np.isfinite(fast[..., :3]).all() == np.isfinite(exact[..., :3]).all()
    lit, _, sigma = literal_statistics(oracle_literal, sc, W, H, frames, 8, 1)
    w0, m0 = compare(fast, ref, sigma, max_sigmas=3.0, max_mean_rel=5e-3)       # against the contract oracle (= parity mode)
    w1, m1 = compare(fast, lit, sigma, max_sigmas=3.0, max_mean_rel=5e-3)       # against the literal oracle
    print(f"perf mode {name}: vs contract worst tile {w0:.2f} sigma, mean {m0:+.2e}; vs literal {w1:.2f} sigma, mean {m1:+.2e}")
