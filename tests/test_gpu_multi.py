"""Several devices behind the C ABI (include/ptmi.h ptmi_multi_*; the caller it serves is the frame loop of
src/renderer/renderer.ts:415-454). A one-GPU box can check two things:

* N = 1 through RCCL: ncclCommInitAll over one device, the frame goes through ncclGather and the unpack kernel and must come
  out as the un-sharded bits;
* the sharding itself for N = 2, 3, 8: N contexts on the one device (PTMI_MULTI_LOOPBACK: device-to-device copies stand in for
  the collective), each rendering its interleaved strips, packed / copied / unpacked by row index — the assembled frame must
  equal the oracle's single render bit for bit, ragged frames and the automatic strip height included.

N > 1 over RCCL has never run from this container (one GPU per box): stated in README.md / DESIGN.md §8."""
import numpy as np
import pytest

from ptmi import layout, shard
from test_gpu_parity import assert_same_floats, bits

pytestmark = pytest.mark.gpu


def test_one_device_through_rccl_gives_the_unsharded_bits(oracle, scene_factory):
    from ptmi import native
    sc = scene_factory("cornell")
    W, H, frames = 96, 70, 4
    cam = layout.make_camera(W, H)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    with native.MultiContext([0]) as m:                         # ncclCommInitAll over one device
        m.upload_scene(sc)
        m.resize(W, H)
        m.set_options(max_bounces=8, do_mis=1)
        m.dispatch(cam, frames)
        m.gather()                                              # pack -> ncclGather -> unpack, all of it on one device
        got = m.read_output()
        st = m.stats()
        assert m.gather_ms() >= 0.0
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, "frame through a one-rank RCCL gather")
        # the loop a preview runs: more frames, another gather, the frame keeps converging on the same buffer
        m.dispatch(layout.make_camera(W, H, frame_index=frames), 2)
        got2 = m.read_output()
    ref2, _ = oracle.render(sc, cam, frames + 2, max_bounces=8, do_mis=1)
    assert_same_floats(got2, ref2, "frames 0..5 through two gathers")


@pytest.mark.parametrize("n,W,H,strip", [(2, 64, 48, 0), (3, 80, 50, 0), (8, 48, 90, 0), (8, 40, 67, 3), (5, 33, 4, 1)])
def test_loopback_shards_assemble_the_oracle_frame(oracle, scene_factory, n, W, H, strip):
    """N contexts on one device, the library's own strip assignment; ragged frames (50 rows over 3 x 4-row strips, 67 rows
    over 8 x 3, 4 rows over 5 devices: one has nothing to render)."""
    from ptmi import native
    sc = scene_factory("cornell")
    frames = 3
    cam = layout.make_camera(W, H)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    with native.MultiContext([0] * n, loopback=True) as m:
        m.upload_scene(sc)
        m.resize(W, H)
        m.set_options(max_bounces=8, do_mis=1, tile_strip=strip, frames_per_batch=2)
        o = m.options()
        assert o.tile_parts == n and o.tile_strip == (strip or shard.strip_rows_for(H, n))
        m.dispatch(cam, 2)
        m.dispatch(layout.make_camera(W, H, frame_index=2), 1)
        got = m.read_output()
        st = m.stats()
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"frame assembled from {n} shards")
        canvas = m.blit()
        assert canvas.shape == (H, W, 4) and canvas[..., 3].min() == 255
        # each context rendered exactly the rows shard.strip_rows names (the contract bench.py's torchrun path relies on too)
        import ctypes
        for r in range(n):
            ctx_r = native.Context.__new__(native.Context)
            ctx_r.L, ctx_r.h, ctx_r.width, ctx_r.height = m.L, ctypes.c_void_p(m.L.ptmi_multi_context(m.h, r)), W, H
            own = ctx_r.read_output()
            ctx_r.h = None                                      # owned by the multi handle
            rows = shard.strip_rows(H, n, r, o.tile_strip)
            if rows:
                assert np.array_equal(bits(own[rows]), bits(ref[rows])), f"rows of shard {r}"
            if r > 0:
                others = sorted(set(range(H)) - set(rows))
                assert not own[others].any(), f"shard {r} wrote rows that are not its own"


def test_multi_errors_are_loud():
    from ptmi import native
    with pytest.raises(native.PtmiError, match="listed twice"):
        native.MultiContext([0, 0])                             # RCCL cannot take one device twice: only the loopback can
    with pytest.raises(native.PtmiError):
        native.MultiContext([0, 99], loopback=True)
    with native.MultiContext([0, 0], loopback=True) as m:
        with pytest.raises(native.PtmiError, match="tile_y0"):
            m.set_options(tile_y0=2, tile_y1=9)
        with pytest.raises(native.PtmiError):
            m.dispatch(layout.make_camera(8, 8), 1)             # no scene yet


def test_resize_to_a_wider_frame_of_the_same_height_reallocates_the_gather_buffers(oracle, scene_factory):
    """the gather's send / receive buffers hold rows_max x W float4 — a resize that keeps the height and widens the
    frame (a horizontal window resize through Renderer.resize) must not keep the smaller ones. 64x48 -> 128x48 on one handle, three
    loopback shards and one device through RCCL; the second frame against the oracle."""
    from ptmi import native
    sc = scene_factory("cornell")
    for devices, loop in (([0, 0, 0], True), ([0], False)):
        with native.MultiContext(devices, loopback=loop) as m:
            m.upload_scene(sc)
            for W, H in ((64, 48), (128, 48), (40, 48)):
                cam = layout.make_camera(W, H)
                m.resize(W, H)
                m.set_options(max_bounces=8, do_mis=1)
                m.dispatch(cam, 2)
                m.gather()
                got = m.read_output()
                ref, _ = oracle.render(sc, cam, 2, max_bounces=8, do_mis=1)
                assert_same_floats(got, ref, f"{W}x{H} after a resize ({len(devices)} shard(s))")


def test_strip_height_cannot_change_while_frames_are_accumulated(scene_factory):
    from ptmi import native
    sc = scene_factory("cornell")
    with native.MultiContext([0, 0], loopback=True) as m:
        m.upload_scene(sc)
        m.resize(64, 48)
        m.set_options(max_bounces=4, do_mis=1, tile_strip=4)
        m.dispatch(layout.make_camera(64, 48), 1)
        with pytest.raises(native.PtmiError):
            m.set_options(tile_strip=2)                         # would hand rows that hold a frame to a device that holds none of them
        m.write_output(np.zeros((48, 64, 4), np.float32))       # every device holds the whole frame again
        m.set_options(tile_strip=2)


def test_two_devices_over_rccl_assemble_the_oracle_frame(oracle, scene_factory):
    """The real N > 1 path: ncclCommInitAll over two devices, one grouped ncclGather with a NULL receive buffer on the non-root,
    unpack of slot 1. Skipped on a one-GPU box (every box this suite has run on so far): the first multi-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (N > 1 over RCCL has never run from the build container)")
    from ptmi import native
    sc = scene_factory("cornell")
    W, H, frames = 96, 70, 3
    cam = layout.make_camera(W, H)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    with native.MultiContext([0, 1]) as m:
        m.upload_scene(sc)
        m.resize(W, H)
        m.set_options(max_bounces=8, do_mis=1)
        m.dispatch(cam, frames)
        got = m.read_output()
        st = m.stats()
    assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
    assert_same_floats(got, ref, "frame gathered from two devices over RCCL")
