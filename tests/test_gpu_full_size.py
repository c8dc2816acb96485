"""BASELINE.json's configurations 1 - 4 on the GPU as written (resolution, frame count, bounces, MIS, depth of field; configs[0]
runs literally in tests/test_gpu_strict.py), checked through what does not depend on the size:
* rows cropped out of the full frame equal the oracle's render of exactly those rows (same camera, all frames), bit for bit;
* the image does not depend on how the work is cut: wavefront batch size, dispatch granularity, interleaved row strips;
* counters are consistent (paths = W*H*frames, per-part segments add up).
The oracle cannot render a whole 1080p x 64 spp frame in test time; eight rows of it take a second."""
import numpy as np
import pytest

from ptmi import layout, native

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool((((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b)))).all())


def render(ctx, cam, frames, **opts):
    ctx.resize(int(cam["width"]), int(cam["height"]))
    base = dict(max_bounces=8, do_mis=1, frames_per_batch=0, tile_y0=0, tile_y1=0, tile_parts=0, tile_part=0, tile_strip=0,
                cull=1, traversal=native.TRAVERSAL_AUTO)
    base.update(opts)
    ctx.set_options(**base)
    ctx.reset_stats()
    ctx.dispatch(cam, frames)
    return ctx.read_output(), ctx.stats()


def test_configs1_cornell_1080p_64spp(gpu_ctx, oracle, scene_factory):
    """BASELINE.json configs[1], the benchmark workload: synthetic Cornell, 1920x1080, frames 0..63, 8 bounces, MIS."""
    sc = scene_factory("cornell")
    W, H, frames = 1920, 1080, 64
    cam = layout.make_camera(W, H)
    gpu_ctx.upload_scene(sc)
    full, st = render(gpu_ctx, cam, frames)
    assert st.paths == W * H * frames and st.frames == frames and st.frames_per_batch_used == 64
    assert st.radiance_stride_bytes == 12                                    # scene in LDS: three floats per path
    assert 2.2 < st.segments / st.paths < 2.5 and 0.5 < st.shadow_rays / st.segments < 0.8
    assert np.isfinite(full).all() and 0.05 < full[..., :3].mean() < 0.5 and full[..., :3].max() <= 2.5

    for y0, y1 in ((0, 3), (537, 543), (1077, 1080)):                       # bottom, middle, top rows vs the oracle
        ref = np.zeros((H, W, 4), np.float32)
        oracle.render(sc, cam, frames, out=ref, y0=y0, y1=y1)
        assert same(full[y0:y1], ref[y0:y1]), f"rows {y0}..{y1} differ from the oracle"

    ragged, st2 = render(gpu_ctx, cam, frames, frames_per_batch=7)          # 9 batches of 7 + 1 of 1
    assert same(ragged, full) and (st2.segments, st2.shadow_rays) == (st.segments, st.shadow_rays)

    gpu_ctx.resize(W, H)                                                    # frame by frame, like renderer.ts:415-454
    gpu_ctx.set_options(frames_per_batch=0)
    for f in range(0, frames, 8):
        gpu_ctx.dispatch(layout.make_camera(W, H, frame_index=f), 8)
    assert same(gpu_ctx.read_output(), full)

    gpu_ctx.resize(W, H)                                                    # four interleaved shares, as four GPUs would
    gpu_ctx.reset_stats()
    for part in range(4):
        gpu_ctx.set_options(tile_parts=4, tile_part=part, tile_strip=4)
        gpu_ctx.dispatch(cam, frames)
    assert same(gpu_ctx.read_output(), full)
    assert gpu_ctx.stats().segments == st.segments and gpu_ctx.stats().paths == st.paths
    gpu_ctx.set_options(tile_parts=0, tile_part=0, tile_strip=0)


def test_configs4_shape_4k_depth_of_field_eight_shares(gpu_ctx, oracle, scene_factory):
    """BASELINE.json configs[4] as written — 3840x2160, all 256 frames, 8 bounces, MIS on, aperture 0.05, focus 2.8, rows
    split into eight shares (what `bench.py --config 4 --gpus 8` gives each rank: 3-row strips) — the eight shares rendered
    one after the other on this one GPU equal the single render, and rows of it equal the oracle."""
    sc = scene_factory("cornell")
    W, H, frames = 3840, 2160, 256
    cam = layout.make_camera(W, H, aperture=0.05, focus_distance=2.8)
    gpu_ctx.upload_scene(sc)
    full, st = render(gpu_ctx, cam, frames)
    assert st.paths == W * H * frames and st.frames_per_batch_used == 16
    for y0, y1 in ((1, 3), (1079, 1082)):
        ref = np.zeros((H, W, 4), np.float32)
        oracle.render(sc, cam, frames, out=ref, y0=y0, y1=y1)
        assert same(full[y0:y1], ref[y0:y1]), f"rows {y0}..{y1} differ from the oracle"
    gpu_ctx.resize(W, H)
    gpu_ctx.reset_stats()
    seg = []
    for part in range(8):
        gpu_ctx.set_options(tile_parts=8, tile_part=part, tile_strip=3)
        before = gpu_ctx.stats().segments
        gpu_ctx.dispatch(cam, frames)
        seg.append(gpu_ctx.stats().segments - before)
    assert same(gpu_ctx.read_output(), full) and sum(seg) == st.segments
    assert max(seg) / (sum(seg) / 8) < 1.02                                 # the shares are balanced to 2 %
    gpu_ctx.set_options(tile_parts=0, tile_part=0, tile_strip=0)


@pytest.mark.parametrize("name,frames,rows", [("cornell_spheres", 512, ((400, 403),)),     # configs[2]: textured PBR spheres, 1024^2 atlas, 512 spp
                                              ("grid_1m", 64, ((300, 302),))])            # configs[3]: 999 708 triangles, depth-29 BVH, 64 spp
def test_configs2_and_3_as_written(gpu_ctx, oracle, scene_factory, name, frames, rows):
    """BASELINE.json configs[2] and configs[3] at their full resolution and frame counts: oracle row crops (the oracle
    renders only those rows, all frames), ragged batches, interleaved shares."""
    sc = scene_factory(name)
    W, H = 1920, 1080
    cam = layout.make_camera(W, H)
    gpu_ctx.upload_scene(sc)
    full, st = render(gpu_ctx, cam, frames)
    # mid-size tree: closest hit from the one-workgroup node cache; 1 M triangles: global memory
    assert st.paths == W * H * frames
    assert st.traversal_used == (native.TRAVERSAL_LDS if name == "cornell_spheres" else native.TRAVERSAL_GLOBAL)
    assert st.radiance_stride_bytes == (12 if name == "cornell_spheres" else 16)     # walked from memory: whole float4 accesses
    assert np.isfinite(full).all() and full[..., :3].mean() > 0.02
    for y0, y1 in rows:
        ref = np.zeros((H, W, 4), np.float32)
        oracle.render(sc, cam, frames, out=ref, y0=y0, y1=y1)
        assert same(full[y0:y1], ref[y0:y1]), f"{name}: rows {y0}..{y1} differ from the oracle"
    ragged, st2 = render(gpu_ctx, cam, frames, frames_per_batch=23)
    assert same(ragged, full) and st2.segments == st.segments
    gpu_ctx.resize(W, H)
    gpu_ctx.reset_stats()
    for part in range(2):
        gpu_ctx.set_options(tile_parts=2, tile_part=part, tile_strip=4)
        gpu_ctx.dispatch(cam, frames)
    assert same(gpu_ctx.read_output(), full) and gpu_ctx.stats().segments == st.segments
    gpu_ctx.set_options(tile_parts=0, tile_part=0, tile_strip=0)
