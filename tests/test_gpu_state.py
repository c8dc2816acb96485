"""GPU parity of ptmi_options.state = 2 (the ray state follows the queue): `shade` of bounce b writes the survivors' origin / RNG
state, direction and throughput at their slot of bounce b's queue into the other of two buffer sets, the next queue lists those
slots, and the radiance stays addressed by the path id each slot carries. Only WHERE the state is kept changes — every RNG
draw, every counter and every radiance bit must equal the oracle's, with the shadow stream on or off, in several batches, with
ray sorting, on two lanes, for 1 ... 8 bounces, from LDS and from global memory.
Reference: the bounce loop of src/shader/pt.wgsl:642-706 (one thread keeps its ray in registers; here it moves between buffers)."""
import numpy as np
import pytest

from ptmi import layout
from test_gpu_parity import assert_same_floats

pytestmark = pytest.mark.gpu


@pytest.fixture()
def st_ctx(gpu_ctx):
    from ptmi import native
    gpu_ctx.set_options(state=2)
    yield gpu_ctx
    gpu_ctx.set_options(state=0, pipeline=0, tails=0, traversal=native.TRAVERSAL_AUTO, cull=1, keep_reference_tree=0, overlap=2, frames_per_batch=0,
                        max_bounces=8, do_mis=1, ray_sort=2, tile_y0=0, tile_y1=0, tile_parts=0)


@pytest.mark.parametrize("name,W,H,frames,bounces,mis,ap", [
    ("cornell", 200, 130, 5, 8, 1, 0.001), ("cornell", 64, 64, 4, 4, 0, 0.001), ("cornell", 96, 64, 3, 1, 1, 0.0),
    ("cornell", 96, 64, 3, 2, 1, 0.0), ("cornell", 96, 64, 3, 3, 1, 0.0), ("cornell_glass", 80, 60, 5, 8, 1, 0.0),
    ("feature_box", 72, 72, 6, 8, 1, 0.05), ("cornell_spheres", 64, 48, 3, 8, 1, 0.0)])
def test_state_follows_the_queue_render_parity(st_ctx, oracle, scene_factory, name, W, H, frames, bounces, mis, ap):
    from ptmi import native
    sc = scene_factory(name)
    cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=bounces, do_mis=mis)
    st_ctx.upload_scene(sc)
    for overlap, fpb, trav, sort in ((1, 0, native.TRAVERSAL_AUTO, 2), (0, 0, native.TRAVERSAL_AUTO, 2), (1, 2, native.TRAVERSAL_GLOBAL, 2),
                                     (1, 0, native.TRAVERSAL_AUTO, 1), (3, 4, native.TRAVERSAL_AUTO, 2)):
        st_ctx.resize(W, H)
        st_ctx.set_options(max_bounces=bounces, do_mis=mis, frames_per_batch=fpb, overlap=overlap, traversal=trav, ray_sort=sort)
        st_ctx.reset_stats()
        st_ctx.dispatch(cam, frames)
        got = st_ctx.read_output()
        st = st_ctx.stats()
        assert st.state_used == 2, "the state did not follow the queue"
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance ({name}, overlap {overlap}, frames_per_batch {fpb}, traversal {trav}, ray_sort {sort})")


def test_state_modes_agree_and_switch_within_one_context(st_ctx, oracle, scene_factory):
    """in place -> following the queue -> in place on one context (the second buffer set is allocated on first use), a resumed
    accumulation across the switch, and interleaved row strips"""
    sc = scene_factory("cornell")
    W, H = 128, 96
    cam = layout.make_camera(W, H, aperture=0.001, focus_distance=5.0)
    ref, _ = oracle.render(sc, cam, 6, max_bounces=8, do_mis=1)
    st_ctx.upload_scene(sc)
    st_ctx.resize(W, H)
    st_ctx.set_options(state=1)
    st_ctx.dispatch(cam, 2)
    assert st_ctx.stats().state_used == 1
    st_ctx.set_options(state=2)
    cam2 = cam.copy(); cam2["frame_index"] = 2
    st_ctx.dispatch(cam2, 2)
    assert st_ctx.stats().state_used == 2
    st_ctx.set_options(state=1, tile_parts=2, tile_part=0, tile_strip=4)
    cam3 = cam.copy(); cam3["frame_index"] = 4
    st_ctx.dispatch(cam3, 2)
    st_ctx.set_options(state=2, tile_parts=2, tile_part=1, tile_strip=4)
    st_ctx.dispatch(cam3, 2)
    assert_same_floats(st_ctx.read_output(), ref, "radiance over three dispatches in alternating state modes")


@pytest.mark.parametrize("seed", range(4))
def test_state_follows_the_queue_random_scene_fuzz(st_ctx, oracle, seed):
    from ptmi import scenes
    sc = scenes.random_soup(60 + seed)
    W, H, frames = 64, 48, 4
    cam = layout.make_camera(W, H, aperture=0.02 if seed % 2 else 0.0, focus_distance=2.5)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    for keep in (0, 1):
        st_ctx.set_options(keep_reference_tree=keep)
        st_ctx.upload_scene(sc)
        st_ctx.resize(W, H)
        st_ctx.reset_stats()
        st_ctx.dispatch(cam, frames)
        got = st_ctx.read_output()
        st = st_ctx.stats()
        assert st.state_used == 2
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance (seed {seed}, keep {keep})")


@pytest.mark.parametrize("name,bounces", [("cornell", 8), ("cornell", 5), ("cornell", 4), ("feature_box", 8)])
def test_serial_tail_is_invisible(st_ctx, oracle, scene_factory, name, bounces):
    """ptmi_options.tails = 2: from bounce 4 on `shadow` runs on the main stream behind its bounce's compaction instead of on
    the side stream beside the next bounce. The radiance is still added by that one kernel in bounce order: same bits, for
    both state modes, several batches, and bounce counts around the switch."""
    sc = scene_factory(name)
    W, H, frames = 96, 64, 5
    cam = layout.make_camera(W, H, aperture=0.001, focus_distance=5.0)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=bounces, do_mis=1)
    st_ctx.upload_scene(sc)
    st_ctx.resize(W, H)
    for state, fpb in ((1, 0), (2, 2)):
        st_ctx.set_options(state=state, tails=2, overlap=1, max_bounces=bounces, do_mis=1, frames_per_batch=fpb)
        st_ctx.reset_stats()
        st_ctx.dispatch(cam, frames)
        got = st_ctx.read_output()
        st = st_ctx.stats()
        assert st.tails_used == (1 if bounces > 4 else 0)
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance ({name}, {bounces} bounces, state {state})")
    st_ctx.set_options(tails=0)


@pytest.mark.parametrize("name,W,H,ap", [("cornell", 96, 64, 0.001), ("feature_box", 72, 72, 0.05)])
def test_raygen_pipeline_is_invisible(st_ctx, oracle, scene_factory, name, W, H, ap):
    """ptmi_options.pipeline = 2: batch k + 1's camera rays are generated on their own stream, into the other of two buffer sets,
    beside batch k. Same kernels, same bits — over many small batches in one dispatch (ragged last batch), over back-to-back
    asynchronous dispatches of different lengths, with the shadow stream on / off, with the state following the queue, and across
    switches of the option."""
    sc = scene_factory(name)
    frames = 11
    cam = layout.make_camera(W, H, aperture=ap, focus_distance=2.8)
    ref, ost = oracle.render(sc, cam, frames, max_bounces=8, do_mis=1)
    st_ctx.upload_scene(sc)
    st_ctx.resize(W, H)
    for overlap, state, fpb in ((1, 1, 2), (0, 1, 3), (1, 2, 2), (1, 1, 0)):
        st_ctx.set_options(pipeline=2, state=state, tails=0, overlap=overlap, max_bounces=8, do_mis=1, frames_per_batch=fpb)
        st_ctx.reset_stats()
        st_ctx.dispatch(cam, frames)                                       # 6 batches of 2 (last: 1), 4 of 3 (last: 2), ..., one of 11
        got = st_ctx.read_output()
        st = st_ctx.stats()
        assert st.pipeline_used == 2
        assert (st.segments, st.shadow_rays, st.paths) == (ost.segments, ost.shadow_rays, ost.paths)
        assert_same_floats(got, ref, f"radiance ({name}, overlap {overlap}, state {state}, frames_per_batch {fpb})")
    # back-to-back dispatches without a synchronisation in between: 1 + 4 + 2 + 3 + 1 frames, the option switched off and on on the way
    st_ctx.set_options(pipeline=2, state=1, overlap=1, frames_per_batch=2)
    k = 0
    for n, pipe in ((1, 2), (4, 2), (2, 1), (3, 2), (1, 2)):
        st_ctx.set_options(pipeline=pipe)
        c2 = cam.copy(); c2["frame_index"] = k
        st_ctx.dispatch(c2, n)
        k += n
    assert k == frames
    assert_same_floats(st_ctx.read_output(), ref, "radiance over five back-to-back dispatches")
    st_ctx.set_options(pipeline=0)
