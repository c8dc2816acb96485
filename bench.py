#!/usr/bin/env python3
"""bench.py — benchmark of the MI355X wavefront path tracer on BASELINE.json's configurations.

    python bench.py [--config {0,1,2,3,4}] --gpus N --steps K --warmup W

--config selects one of BASELINE.json's `configs` (default 1, the configuration the metric is quoted on):

    0  Cornell,            256x256,  16 spp, 4 bounces, MIS off   (configs[0]; on the GPU here — the reference has no CPU path)
    1  Cornell,          1920x1080,  64 spp, 8 bounces, MIS on    (configs[1])
    2  Cornell + spheres, 1920x1080, 512 spp, textured PBR        (configs[2])
    3  1 M-triangle grid, 1920x1080,  64 spp                      (configs[3]; the only scene that reaches the memory side)
    4  Cornell,          3840x2160, 256 spp, depth of field on (aperture 0.05, focus 2.8)  (configs[4])

One step = one ptmi_dispatch of --frames-per-step frames (default 64, traced as wavefront batches of up to 128 Mi
paths: at 1920x1080 one batch of 133 M paths) over the rank's rows; K defaults to spp / frames-per-step, so the default
run renders exactly the config's spp (config 1: ONE step of 64 frames after one warm-up step). Scene and output live in HBM before the timed region starts (the C ABI copies host blobs at upload).

N > 1 (launched by torch.distributed.run, one rank per GPU) shards pixel rows, with no data-path collective (pixels
and RNG streams are independent, pt.wgsl:719, :753-761); after the last step the strips are gathered to rank 0 with
ONE RCCL gather, inside the timed region. value = all ranks' path segments / max-over-ranks time.
  configs 0-3: WEAK scaling — the same view at N times the pixels (1920x1080, 2720x1528, 3840x2160, 5424x3040: same
               aspect, so the same mix of cheap and expensive pixels per rank).
  config 4:    STRONG scaling — the 3840x2160 frame is fixed and split over the ranks (BASELINE.json: "pixel-tile shard
               across 8x MI355X with RCCL gather").
Rank r renders the strips r, r + N, ... (4 rows each, fewer when the frame height is not a whole number of such rounds: 3 for 2160 rows over 8 ranks).

Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel against HBM — the kernel with the most device time among
extend, shade and shadow, whatever stream it runs on (on the Cornell scenes the any-hit traversal `shadow`: 72 B per traced
record — index 4 + record 44 read, radiance read-modify-write 24; on the 1 M-triangle scene the closest-hit traversal `extend`,
44 B per ray) — with launch durations from HIP events recorded by the library around every launch on the stream
it runs on (`--timing 3`: every kernel); `roofline.kernels` carries the same figures for all three kernels, `pipeline_frac` the
whole dispatch, `valu_issue` the second roofline (vector-ALU issue) with the lane utilisation of each kernel.
`roofline.traffic` and `roofline.valu_issue` replay the committed rocprofv3 counter passes of the same command
(profiles/<tag>_cfgN_pmc.json / _counters.json — counters cannot be read from inside an un-profiled run); profiles/<tag>_manifest.json
names the commit and the hash of the kernel sources they were measured on, and both carry `stale: true` when the sources of
this build differ.
`cpu_baseline` times the CPU oracle (oracle/, a restatement — the reference has no CPU path) on a bounded sample of
the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

EXTEND_BYTES_PER_RAY = 32 + 4 + 8       # queue 4 + O,D 32 read, hit record (t, triangle) 8 written
SHADOW_RECORD_BYTES = 44                # origin + distance 16, direction + path id 16, contribution 12


def shadow_bytes_per_ray(l_bytes=12):
    """index 4 + record 44 read, radiance read-modify-write (unoccluded): 2 x 12, or 2 x 16 where the library stores the
    per-path radiance at 16-byte stride (scenes walked from memory; ptmi_stats.radiance_stride_bytes)."""
    return 4 + SHADOW_RECORD_BYTES + 2 * l_bytes
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
PROFILE_TAG = "r04"                     # profiles/<tag>_cfgN_*.json are the counter passes replayed in `roofline`

# kernel names in the counter files (tools/pmc_summary.py): traverse.hip's and traverse_own.hip's; a config runs one per kind
EXTEND_KEYS = ("k_trace_lds/extend", "k_trace_global/extend", "k_own_lds/extend", "k_own_global/extend")
SHADOW_KEYS = ("k_trace_lds/shadow", "k_trace_global/shadow", "k_own_lds/shadow", "k_own_global/shadow")

CONFIGS = {
    0: dict(scene="cornell", width=256, height=256, spp=16, fps=16, bounces=4, mis=0, aperture=0.001, focus=5.0,
            scaling="weak", name="configs[0]"),
    1: dict(scene="cornell", width=1920, height=1080, spp=64, fps=64, bounces=8, mis=1, aperture=0.001, focus=5.0,
            scaling="weak", name="configs[1]"),
    2: dict(scene="cornell_spheres", width=1920, height=1080, spp=512, fps=64, bounces=8, mis=1, aperture=0.001, focus=5.0,
            scaling="weak", name="configs[2]"),
    3: dict(scene="grid_1m", width=1920, height=1080, spp=64, fps=64, bounces=8, mis=1, aperture=0.001, focus=5.0,
            scaling="weak", name="configs[3]"),
    4: dict(scene="cornell", width=3840, height=2160, spp=256, fps=64, bounces=8, mis=1, aperture=0.05, focus=2.8,
            scaling="strong", name="configs[4]"),
}


def shade_bytes_per_segment(do_mis, p_record, bounce0_share):
    """queue 4 + hit 8 + O,D,C 40 read, O,D,C 40 written (survivors; priced for every segment), ballots 1/4,
    + 44 per emitted shadow record; bounce 0 reads neither a queue nor a stored throughput (C)."""
    b = 4 + 8 + 40 + 40 + 0.25 + (SHADOW_RECORD_BYTES * p_record if do_mis else 0.0)
    return b - bounce0_share * (4 + 8)


def pipeline_bytes_per_segment(do_mis, mean_len, l_bytes=12):
    """SURVEY.md §8(d) re-derived for this build's records:
    extend R 32+4 W 8; shade R 4+8+40 (O,D,C) W 40 + masks 0.25; compact R 4 W 4;
    MIS: shadow record W 44 R 44 + radiance RMW 2 x l_bytes; per path: raygen W 32 + l_bytes (O, D, L), accumulate
    R l_bytes + frame RMW 32, less what bounce 0 does not touch (the identity queue, 3 x 4, and the stored part of the
    throughput, 8). l_bytes: 12, or 16 for scenes walked from memory."""
    seg = (32 + 4 + 8) + (4 + 8 + 40 + 40) + 8
    if do_mis:
        seg += 2 * SHADOW_RECORD_BYTES + 2 * l_bytes
    return seg + ((32 + l_bytes) + (l_bytes + 32) - 12 - 8) / max(mean_len, 1e-9)


def csrc_sha():
    """sha256 over the kernel sources (wgpu-path-tracing_amd/csrc/*.hip, *.h, include/*.h, file names included): what a
    committed counter file must have been measured on for its replay to describe THIS build."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "wgpu-path-tracing_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def manifest():
    """profiles/<tag>_manifest.json (tools/make_manifest.py): commit and source hash of the build the counter files of this tag
    were measured on, and the sha256 of every file — the GPU box has no git history to ask."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_manifest.json")
    try:
        return json.load(open(path))
    except Exception:
        return {}


def profile_path(config, kind):
    return os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_cfg{config}_{kind}.json")


def pmc_traffic(config, is_profiled_workload):
    """HBM bytes per launch of the traversal and shade kernels from the committed rocprofv3 PMC passes of this exact
    command (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, KB per launch; gfx950 tallies wide reads at half:
    traffic = 2*FETCH_SIZE + WRITE_SIZE). {} when the workload differs from the profiled one."""
    path = profile_path(config, "pmc")
    if not is_profiled_workload or not os.path.exists(path):
        return {}, None
    d = json.load(open(path))
    out = {}
    for label, keys in (("extend", EXTEND_KEYS), ("shadow", SHADOW_KEYS), ("shade", ("k_shade",))):
        f = w = 0.0
        for k in keys:          # a config uses one variant per kernel; a missing key contributes nothing
            f += d.get("FETCH_SIZE", {}).get(k, {}).get("avg_KB_per_launch", 0.0) * d.get("FETCH_SIZE", {}).get(k, {}).get("launches", 0)
            w += d.get("WRITE_SIZE", {}).get(k, {}).get("avg_KB_per_launch", 0.0) * d.get("WRITE_SIZE", {}).get(k, {}).get("launches", 0)
        n = sum(d.get("FETCH_SIZE", {}).get(k, {}).get("launches", 0) for k in keys)
        if n:
            out[label] = int((2 * f + w) / n * 1024)
    man = manifest()
    src = {"file": os.path.relpath(path, ROOT), "file_commit": man.get("commit"), "code_commit": d.get("_code_commit"),
           "manifest": f"profiles/{PROFILE_TAG}_manifest.json" if man else None,
           # replayed, not measured in this run: stale = the kernel sources have changed since the counters were taken
           "stale": man.get("csrc_sha") != csrc_sha(),
           "formula": "2*FETCH_SIZE + WRITE_SIZE, bytes per launch (mean over the launches of the profiled run: this config's "
                      "dispatch, one warm-up and one timed step)"}
    return out, src


VALU_CLOCK_GHZ = 2.4                    # MI355X_MICROARCH.md: peak engine clock; the chip runs lower under load, so the fraction is a floor
N_SIMDS = 256 * 4


def valu_issue(config, is_profiled_workload, launches, gpu_ms):
    """The bound the traversal kernels (and, with them beside it, the whole dispatch) actually run against: vector-ALU issue.
    SQ_ACTIVE_INST_VALU of the committed counter pass (profiles/<tag>_cfgN_counters.json; quad-cycles, summed over all SIMDs,
    mean per launch) x launches of this step = SIMD-cycles the vector ALUs were busy; / (1024 SIMDs x 2.4 GHz x device time).
    None when there is no counter file for this workload."""
    path = profile_path(config, "counters")
    if not is_profiled_workload or not os.path.exists(path) or gpu_ms <= 0:
        return None
    d = json.load(open(path)).get("SQ_ACTIVE_INST_VALU", {})
    per = {}
    for label, keys in (("extend", EXTEND_KEYS), ("shadow", SHADOW_KEYS), ("shade", ("k_shade",))):
        q = sum(d.get(k, {}).get("avg_per_launch", 0.0) for k in keys)
        if q:
            per[label] = q
    if len(per) < 3:
        return None
    # lane utilisation of VALU instructions from the same pass: thread-cycles / (64 x instruction quad-cycles)
    tc = json.load(open(path)).get("SQ_THREAD_CYCLES_VALU", {})
    lane_util = {}
    for label, keys in (("extend", EXTEND_KEYS), ("shadow", SHADOW_KEYS), ("shade", ("k_shade",))):
        t = sum(tc.get(k, {}).get("avg_per_launch", 0.0) for k in keys)
        if t and per.get(label):
            lane_util[label] = round(t / (64.0 * per[label]), 4)
    busy_ms = {k: 4.0 * q * launches.get(k, 0) / N_SIMDS / (VALU_CLOCK_GHZ * 1e6) for k, q in per.items()}
    total = sum(busy_ms.values())
    return {"bound": "valu issue", "busy_ms_per_step_at_peak_clock": {k: round(v, 3) for k, v in busy_ms.items()},
            "busy_ms_total": round(total, 3), "device_ms": round(gpu_ms, 3), "frac": round(total / gpu_ms, 4),
            "clock_ghz": VALU_CLOCK_GHZ, "simds": N_SIMDS, "lane_utilisation": lane_util,
            "stale": manifest().get("csrc_sha") != csrc_sha(),
            "source": {"file": os.path.relpath(path, ROOT), "counter": "SQ_ACTIVE_INST_VALU (quad-cycles per launch, all SIMDs)",
                       "code_commit": json.load(open(path)).get("_code_commit")},
            "note": "replayed from the committed counter pass of this command; extend, shade and shadow only (raygen, compaction and "
                    "accumulate are streaming kernels). busy_ms are issue times, not wall times: shadow runs on its own stream beside "
                    "extend / shade, so the three kernels' event times add up to more than device_ms while their issue times cannot "
                    "exceed it; the clock under load is below the peak used here, so the true fraction is higher"}


def available_cores():
    """The cores this process may actually use: the host's count, cut down to the scheduler affinity and to the cgroup's CPU quota (a
    GPU box of the pool shows 256 host cores to a container that is allowed 16: 256 OpenMP threads on 16 cores' worth of time run at
    15 Msamples/s where 16 threads reach 25). Returns (usable, host)."""
    host = os.cpu_count() or 1
    n = host
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, n), host


def cpu_baseline(scene, cam_kw, width, height, bounces, mis, threads, target_s=12.0):
    """Oracle on whole frames of the same camera: 1 calibration frame, then as many frames as
    fit in about target_s seconds (at most the 64 of the workload)."""
    from oracle_lib import Oracle
    from ptmi import layout
    import numpy as np
    orc = Oracle(strict=False)
    out = np.zeros((height, width, 4), np.float32)
    # one calibration frame on all usable cores — and, where a container is shown more cores than it is given time on (no quota to
    # read), on 16 threads as well: the sample runs with whichever was faster (`cores` says which)
    candidates = [threads] + ([16] if threads > 32 else [])
    best = None
    for th in candidates:
        _, st = orc.render(scene, layout.make_camera(width, height, **cam_kw), 1, max_bounces=bounces, do_mis=mis, out=out, threads=th)
        if best is None or st.seconds < best[1].seconds:
            best = (th, st)
    threads, st0 = best
    frames = int(max(1, min(63, target_s / max(st0.seconds, 1e-3))))
    _, st = orc.render(scene, layout.make_camera(width, height, frame_index=1, **cam_kw), frames, max_bounces=bounces,
                       do_mis=mis, out=out, threads=threads)
    return {
        "value": round(st.segments / st.seconds / 1e6, 4), "unit": "Msamples/s", "cores": int(st.threads),
        "cores_available": available_cores()[0], "cores_host": available_cores()[1],
        "kind": "port",
        "sample": f"frames 1..{frames} of the full {width}x{height} frame: {st.paths} paths, {st.segments} segments "
                  f"in {st.seconds:.2f} s (oracle/pt_oracle.c contract build, OpenMP dynamic rows)",
    }


def single_process(args, cfg, overridden, steps, fps, spp):
    """--single-process: N devices driven by ONE process through ptmi_multi_* — the reference's own model (one host thread, one
    Renderer: src/renderer/renderer.ts:415-454). Same sharding (interleaved strips), the library's own pack -> ncclGather -> unpack
    inside the timed region, the same JSON line (per-kernel times are the maximum over the devices). With --rehearse every "device"
    is GPU 0 and device-to-device copies stand in for the collective (PTMI_MULTI_LOOPBACK): the packing, not a benchmark."""
    import numpy as np
    from ptmi import layout, native, scenes, shard
    n = args.gpus
    strong = cfg["scaling"] == "strong"
    W, H = (cfg["width"], cfg["height"]) if strong else shard.weak_frame(cfg["width"], cfg["height"], n)
    cam_kw = dict(aperture=cfg["aperture"], focus_distance=cfg["focus"])
    scene = scenes.make(cfg["scene"])
    m = native.MultiContext([0] * n if args.rehearse else list(range(n)), loopback=bool(args.rehearse))
    upload_opts = {}
    for k, v in (("tree_builder", args.tree_builder), ("leaves", args.leaves), ("leaf_tris", args.leaf_tris)):
        if v is not None:
            upload_opts[k] = v
    if upload_opts:
        m.set_options(**upload_opts)
    t_up = time.perf_counter()
    m.upload_scene(scene)
    upload_wall_ms = (time.perf_counter() - t_up) * 1e3
    m.resize(W, H)
    trav = {"auto": native.TRAVERSAL_AUTO, "global": native.TRAVERSAL_GLOBAL, "lds": native.TRAVERSAL_LDS,
            "global_exact": native.TRAVERSAL_GLOBAL_EXACT}[args.traversal]
    m.set_options(max_bounces=cfg["bounces"], do_mis=cfg["mis"], frames_per_batch=args.frames_per_batch, traversal=trav, cull=1, timing=args.timing,
                  **({"overlap": args.overlap} if args.overlap is not None else {}))
    frame_index = 0
    enqueue_ms = []

    def step():
        nonlocal frame_index
        t = time.perf_counter()
        m.dispatch(layout.make_camera(W, H, frame_index=frame_index, **cam_kw), fps)
        enqueue_ms.append((time.perf_counter() - t) * 1e3)      # host time of enqueuing N devices from one thread
        frame_index += fps

    for _ in range(args.warmup):
        step()
    m.gather()
    m.synchronize()
    m.reset_stats()
    enqueue_ms.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    m.gather()
    m.synchronize()
    dt = time.perf_counter() - t0
    st = m.stats()
    ok = None
    if args.rehearse:                # the gathered frame against one device's unsharded render of the same frames
        got = m.read_output()
        with native.Context(0) as c:
            c.upload_scene(scene)
            c.resize(W, H)
            c.set_options(max_bounces=cfg["bounces"], do_mis=cfg["mis"])
            for k in range(args.warmup + steps):
                c.dispatch(layout.make_camera(W, H, frame_index=k * fps, **cam_kw), fps)
            ok = bool(np.array_equal(got.view(np.uint32), c.read_output().view(np.uint32)))
    segments, paths = float(st.segments), float(st.paths)
    msamples = segments / dt / 1e6
    out = {
        "metric": "Msamples/s (rays x bounces / s)", "value": round(msamples, 3), "unit": "Msamples/s",
        "n_gpus": n, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
        "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE.json {cfg['name']}: {cfg['scene']} "
                        + (f"{W}x{H} frame split over {n} GPU(s)" if strong else f"{cfg['width']}x{cfg['height']} pixels per GPU (frame {W}x{H})")
                        + f", {spp} spp, {cfg['bounces']} bounces, MIS {'on' if cfg['mis'] else 'off'}, aperture {cfg['aperture']:g}, focus {cfg['focus']:g}"
                        + (f"; overridden: {','.join(overridden)}" if overridden else ""),
            "config_index": args.config, "frames_per_step": fps, "frames_per_batch": int(st.frames_per_batch_used),
            "driver": "one process, ptmi_multi_* (one host thread, one stream per device, ncclGather behind the C ABI)",
            "parallelism": f"{int(m.options().tile_strip)}-row strips x{n}", "leaves": int(st.leaves_used),
            "extend_variant": int(st.extend_variant), "shadow_variant": int(st.shadow_variant),
        },
        **({"rehearsal": {"sharded_equals_unsharded_bitwise": ok, "backend": "loopback copies on one GPU", "note": "not a benchmark"}} if args.rehearse else {}),
        "segments": int(segments), "shadow_rays": int(st.shadow_rays), "paths": int(paths), "mean_path_length": round(segments / max(paths, 1), 4),
        "gpu_ms_max_over_devices": round(st.gpu_ms, 3),
        "kernel_ms_max_over_devices": {"extend": round(st.extend_ms, 3), "shade": round(st.shade_ms, 3), "shadow": round(st.shadow_ms, 3),
                                       "raygen": round(st.raygen_ms, 3), "compact": round(st.compact_ms, 3), "accumulate": round(st.accumulate_ms, 3)},
        "gather_ms": round(m.gather_ms(), 4),
        # one thread enqueues every device's kernels in turn: device i starts this much after device 0 (the launch skew of a step)
        "enqueue_ms_per_step": {"mean": round(float(np.mean(enqueue_ms)), 3), "max": round(float(np.max(enqueue_ms)), 3),
                                "per_device_mean": round(float(np.mean(enqueue_ms)) / n, 3)},
        "upload_ms": {"wall_all_devices": round(upload_wall_ms, 2), "library_max": round(st.upload_ms, 2)},
        "roofline": {"bound": "hbm", "kernel": "pipeline (whole dispatch)", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "achieved": round(msamples * 1e6 * pipeline_bytes_per_segment(cfg["mis"], segments / max(paths, 1), int(st.radiance_stride_bytes) or 12) / 1e9, 3),
                     "frac": round(msamples * 1e6 * pipeline_bytes_per_segment(cfg["mis"], segments / max(paths, 1), int(st.radiance_stride_bytes) or 12) / 1e9 / HBM_PEAK_GBS, 6),
                     "traffic": None},
    }
    print(json.dumps(out), flush=True)
    m.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: the config's spp / frames-per-step")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None, help="rows per GPU (weak scaling) or of the whole frame (config 4)")
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--no-mis", action="store_true")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--aperture", type=float, default=None)
    ap.add_argument("--focus-distance", type=float, default=None)
    ap.add_argument("--frames-per-batch", type=int, default=0)
    ap.add_argument("--traversal", default="auto", choices=["auto", "global", "lds", "global_exact"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--perf-mode", type=int, default=0, help="library option perf_mode (0 = parity arithmetic, the headline)")
    ap.add_argument("--overlap", type=int, default=None,
                    help="library option overlap (0: one stream; 1: shadow kernel on a second stream; default: the library's)")
    ap.add_argument("--tree-builder", type=int, default=None, help="library option tree_builder (1: host SAH, 2: GPU linear BVH), read at upload")
    ap.add_argument("--leaves", type=int, default=None,
                    help="library option leaves, read at upload (1: the uploaded BVH's leaves; 2: the library's own leaves; default: the library's)")
    ap.add_argument("--leaf-tris", type=int, default=None, help="library option leaf_tris (most triangles per own leaf)")
    ap.add_argument("--no-leaves-compare", action="store_true",
                    help="skip the second timed leg (N = 1 only) that renders the same steps with the OTHER leaf mode for `leaves_compare`")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 from ONE process: ptmi_multi_* (one host thread, one stream per device, the library's own RCCL gather) "
                         "instead of one rank per GPU under torch.distributed")
    ap.add_argument("--keep-reference-tree", action="store_true",
                    help="walk the BVH exactly as uploaded instead of the hierarchy rebuilt over its leaves")
    ap.add_argument("--rehearse", action="store_true",
                    help="dry run of the N>1 path on ONE GPU: every rank uses device 0, gloo backend, strips gathered "
                         "through host memory, rank 0 checks the gathered frame bit for bit against its own unsharded "
                         "render. Not a benchmark.")
    ap.add_argument("--timing", type=int, default=3, help="library HIP-event timing level (2 = every extend launch, 3 = every kernel)")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    overridden = []
    for key, val in (("scene", args.scene), ("width", args.width), ("height", args.height), ("bounces", args.bounces),
                     ("fps", args.frames_per_step), ("aperture", args.aperture), ("focus", args.focus_distance)):
        if val is not None and val != cfg[key]:
            cfg[key] = val
            overridden.append(key)
    if args.no_mis and cfg["mis"]:
        cfg["mis"] = 0
        overridden.append("mis")
    fps = cfg["fps"]
    steps = args.steps if args.steps is not None else max(1, cfg["spp"] // fps)
    spp = steps * fps

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        return single_process(args, cfg, overridden, steps, fps, spp)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # not under a launcher: start one rank per GPU ourselves — fresh child processes, before this one has touched a GPU —
            # and pass rank 0's JSON line through (the driver's own command is exactly this child command)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
        args.gpus = world

    import numpy as np
    import torch
    from ptmi import layout, native, scenes, shard

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the path tracer has no CPU backend")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mis = cfg["mis"]
    strong = cfg["scaling"] == "strong"
    if strong:
        W, H = cfg["width"], cfg["height"]                     # the frame is fixed; ranks split it
    else:
        # the SAME view at `world` times the pixels, so the mix of cheap and expensive pixels does not change with N
        W, H = shard.weak_frame(cfg["width"], cfg["height"], world)
    strip = shard.strip_rows_for(H, world)
    cam_kw = dict(aperture=cfg["aperture"], focus_distance=cfg["focus"])
    scene = scenes.make(cfg["scene"])

    ctx = native.Context(local_rank)
    upload_opts = {"keep_reference_tree": int(args.keep_reference_tree)}
    for k, v in (("tree_builder", args.tree_builder), ("leaves", args.leaves), ("leaf_tris", args.leaf_tris)):
        if v is not None:
            upload_opts[k] = v
    ctx.set_options(**upload_opts)
    t_up = time.perf_counter()
    ctx.upload_scene(scene)
    upload_wall_ms = (time.perf_counter() - t_up) * 1e3
    ctx.resize(W, H)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")       # binding 0, owned by the caller
    ctx.bind_output_device(frame.data_ptr(), frame.numel() * 4)
    # One explicit stream for rendering, copies and the collective. (torch's default stream has the NULL
    # handle, which ptmi_set_stream reads as "use the context's own stream"; that stream is not ordered
    # against torch work, so a gather could overtake the render.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    trav = {"auto": native.TRAVERSAL_AUTO, "global": native.TRAVERSAL_GLOBAL, "lds": native.TRAVERSAL_LDS,
            "global_exact": native.TRAVERSAL_GLOBAL_EXACT}[args.traversal]
    extra = {}
    if args.perf_mode:
        extra["perf_mode"] = args.perf_mode
    if args.overlap is not None:
        extra["overlap"] = args.overlap
    ctx.set_options(max_bounces=cfg["bounces"], do_mis=mis, frames_per_batch=args.frames_per_batch, traversal=trav, cull=1,
                    timing=args.timing, **extra, **shard.strip_options(world, rank, strip))

    frame_index = 0

    def step():
        nonlocal frame_index
        ctx.dispatch(layout.make_camera(W, H, frame_index=frame_index, **cam_kw), fps)
        frame_index += fps

    # row indices, the packed send buffer and the root's receive buffers exist before any timed step (shard.StripGather)
    strip_gather = shard.StripGather(frame, world, rank, strip) if world > 1 and not args.rehearse else None

    def gather():
        # the strips accumulate locally; ONE gather assembles the frame after the last frame
        if world > 1 and args.rehearse:
            host = frame.cpu()
            shard.gather_strips(dist, host, world, rank, strip)
            if rank == 0:
                frame.copy_(host)
        elif world > 1:
            strip_gather.run(dist)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_leg = time.perf_counter()
    for _ in range(args.warmup):
        step()
    gather()                                  # also sets up the RCCL channels outside the timed region
    fence()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    gather()
    fence()
    dt = time.perf_counter() - t0
    leg_seconds = {"gpu_warmup_and_timed": round(time.perf_counter() - t_leg, 3)}

    st = ctx.stats()
    # the same steps with the OTHER leaf mode (N = 1): both numbers stay in the record while the own leaves are new
    leaves_compare = None
    if world == 1 and not args.no_leaves_compare and not args.rehearse and not args.keep_reference_tree:
        t_leg = time.perf_counter()
        other = 1 if int(st.leaves_used) == 2 else 2
        ctx.set_options(leaves=other)
        ctx.upload_scene(scene)
        frame_index = 0
        for _ in range(args.warmup):
            step()
        fence()
        ctx.reset_stats()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt2 = time.perf_counter() - t1
        st2 = ctx.stats()
        leaves_compare = {"leaves": int(st2.leaves_used), "value": round(st2.segments / dt2 / 1e6, 3), "ms_per_step": round(dt2 / steps * 1e3, 3),
                          "segments": int(st2.segments), "same_segments": bool(int(st2.segments) == int(st.segments)),
                          "extend_variant": int(st2.extend_variant), "shadow_variant": int(st2.shadow_variant),
                          "kernel_ms": {"extend": round(st2.extend_ms, 3), "shade": round(st2.shade_ms, 3), "shadow": round(st2.shadow_ms, 3)}}
        leg_seconds["gpu_leaves_compare"] = round(time.perf_counter() - t_leg, 3)
    rehearsal_ok = None
    if args.rehearse and rank == 0:
        # the same frames, unsharded, on this rank alone: the sharded + gathered frame must equal it bit for bit
        gathered = frame.cpu().numpy().copy()
        ctx.set_options(tile_y0=0, tile_y1=0, tile_parts=0, tile_part=0, timing=0)
        frame.zero_()
        for k in range(args.warmup + steps):
            ctx.dispatch(layout.make_camera(W, H, frame_index=k * fps, **cam_kw), fps)
        torch.cuda.synchronize()
        rehearsal_ok = bool(np.array_equal(gathered.view(np.uint32), frame.cpu().numpy().view(np.uint32)))
    t = torch.tensor([dt, float(st.segments), float(st.shadow_rays), float(st.paths)], dtype=torch.float64,
                     device="cpu" if args.rehearse else "cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    segments, shadow_rays, paths = float(t[1]), float(t[2]), float(t[3])

    if rank == 0:
        mean_len = segments / paths
        msamples = segments / dt / 1e6
        l_bytes = int(st.radiance_stride_bytes) or 12
        b_seg = pipeline_bytes_per_segment(mis, mean_len, l_bytes)
        # the counter files hold per-launch means of this config's dispatch; every step is such a dispatch (64 more frames of the
        # same view), so they apply whatever --steps / --warmup are; any flag that changes the dispatch itself rules them out
        is_profiled = (not overridden and world == 1 and args.traversal == "auto"
                       and not args.perf_mode and args.overlap is None and not args.keep_reference_tree
                       and args.frames_per_batch == 0 and args.tree_builder is None and args.leaves is None and args.leaf_tris is None)
        traffic, traffic_src = pmc_traffic(args.config, is_profiled)

        def kernel_entry(label, name, ms, launches, units, bytes_per_unit):
            if not launches or ms <= 0:
                return None
            gbs = units * bytes_per_unit / 1e9 / (ms / 1e3)
            e = {"kernel": name, "bytes_per_unit": round(bytes_per_unit, 2), "units_per_launch": round(units / launches, 1),
                 "algorithmic_bytes_per_launch": int(units * bytes_per_unit / launches),
                 "avg_launch_ms": round(ms / launches, 4), "launches": int(launches),
                 "achieved": round(gbs, 3), "frac": round(gbs / HBM_PEAK_GBS, 6), "traffic": traffic.get(label)}
            if traffic.get(label):
                e["traffic_gbs"] = round(traffic[label] / 1e9 / (ms / launches / 1e3), 3)
                e["traffic_frac"] = round(e["traffic_gbs"] / HBM_PEAK_GBS, 6)
            return e

        p_record = st.shadow_traced / max(st.segments, 1)
        b0_share = (st.segments_by_bounce[0] / st.segments) if st.segments else 0.0
        ext = kernel_entry("extend", "extend (closest-hit BVH traversal)", st.extend_ms, st.extend_launches, st.segments,
                           EXTEND_BYTES_PER_RAY)
        shd = kernel_entry("shade", "shade (material, next-event record, BSDF sample, Russian roulette)", st.shade_ms,
                           st.shade_launches, st.segments, shade_bytes_per_segment(mis, p_record, b0_share))
        shw = kernel_entry("shadow", "shadow (any-hit visibility of the next-event record)", st.shadow_ms, st.shadow_launches,
                           st.shadow_traced, shadow_bytes_per_ray(l_bytes))
        # The dominant kernel = the one with the most device time among extend, shade and shadow, whatever stream it ran on
        # (HIP events on the stream each kernel is launched on; with the shadow kernel on its own stream the three times add
        # up to more than the dispatch time — see kernel_ms_sum_over_gpu_ms)
        overlapped = bool(mis and (args.overlap is None or args.overlap != 0))
        cands = [(st.extend_ms, ext), (st.shade_ms, shd), (st.shadow_ms, shw)]
        dom = max((c for c in cands if c[1]), key=lambda c: c[0], default=(0, None))[1]
        dominant_by = "most device time among extend, shade and shadow (HIP events on the streams they run on)"
        kernel_ms = {"extend": st.extend_ms, "shade": st.shade_ms, "shadow": st.shadow_ms, "raygen": st.raygen_ms,
                     "compact": st.compact_ms, "accumulate": st.accumulate_ms}
        par = f"{strip}-row strips x{world}"
        out = {
            "metric": "Msamples/s (rays x bounces / s)", "value": round(msamples, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": cfg["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"BASELINE.json {cfg['name']}: {cfg['scene']} "
                            + (f"{W}x{H} frame split over {world} GPU(s)" if strong else f"{cfg['width']}x{cfg['height']} pixels per GPU (frame {W}x{H})")
                            + f", {spp} spp, {cfg['bounces']} bounces, MIS {'on' if mis else 'off'}, aperture {cfg['aperture']:g}, "
                              f"focus {cfg['focus']:g}" + (f"; overridden: {','.join(overridden)}" if overridden else ""),
                "config_index": args.config,
                "frames_per_step": fps, "frames_per_batch": int(st.frames_per_batch_used),
                "traversal": "lds" if st.traversal_used == native.TRAVERSAL_LDS else "global",
                "triangles": int(len(scene.tris)), "bvh_nodes": int(len(scene.nodes)), "parallelism": par,
                **({"perf_mode": args.perf_mode} if args.perf_mode else {}),
                "leaves": int(st.leaves_used), "leaf_tris": int(st.leaf_tris_used),
                "extend_variant": int(st.extend_variant), "shadow_variant": int(st.shadow_variant),
            },
            "verify_failed_rank0": int(st.verify_failed),
            **({"leaves_compare": leaves_compare} if leaves_compare else {}),
            **({"rehearsal": {"sharded_equals_unsharded_bitwise": rehearsal_ok, "backend": "gloo", "note": "all ranks on one GPU; not a benchmark"}} if args.rehearse else {}),
            "segments": int(segments), "shadow_rays": int(shadow_rays), "paths": int(paths),
            "shadow_traced_rank0": int(st.shadow_traced),
            "mean_path_length": round(mean_len, 4),
            "segments_by_bounce_rank0": [int(v) for v in list(st.segments_by_bounce)[:cfg["bounces"]]],
            "nominal_msamples": round(paths * cfg["bounces"] / dt / 1e6, 3),
            "gpu_ms_rank0": round(st.gpu_ms, 3),
            "kernel_ms_rank0": {k: round(v, 3) for k, v in kernel_ms.items()},
            # with the shadow kernel on its own stream (library option overlap, the default with next-event estimation) kernels
            # run beside each other: their HIP-event times then add up to MORE than the dispatch time; --overlap 0 puts
            # everything on one stream, where the sum must equal it
            "shadow_overlapped": overlapped,
            "kernel_ms_sum_over_gpu_ms": round(sum(kernel_ms.values()) / st.gpu_ms, 4) if st.gpu_ms > 0 else None,
            "upload_ms_rank0": {"wall": round(upload_wall_ms, 2), "library": round(st.upload_ms, 2),
                                "rebuilt_hierarchy": round(st.upload_tree_ms, 2), "copies": round(st.upload_copy_ms, 2)},
            "roofline": {
                "bound": "hbm", "kernel": dom["kernel"] if dom else None, "dominant_by": dominant_by,
                "achieved": dom["achieved"] if dom else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"] if dom else None,
                "traffic": dom["traffic"] if dom else None,
                "traffic_source": traffic_src,
                **({k: dom[k] for k in ("algorithmic_bytes_per_launch", "bytes_per_unit", "units_per_launch", "avg_launch_ms",
                                        "launches")} if dom else {}),
                **({k: dom[k] for k in ("traffic_gbs", "traffic_frac") if k in dom} if dom else {}),
                "kernels": {k: v for k, v in (("extend", ext), ("shade", shd), ("shadow", shw)) if v},
                "pipeline_bytes_per_segment": round(b_seg, 1),
                "pipeline_achieved": round(msamples * 1e6 * b_seg / 1e9, 3),
                "pipeline_frac": round(msamples * 1e6 * b_seg / 1e9 / HBM_PEAK_GBS, 6),
                # the second roofline: how busy the vector ALUs were (counter replay; None without a counter file)
                "valu_issue": valu_issue(args.config, is_profiled, {"extend": st.extend_launches, "shade": st.shade_launches,
                                                                  "shadow": st.shadow_launches}, st.gpu_ms),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            t_leg = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(scene, cam_kw, W, H, cfg["bounces"], mis, available_cores()[0])
            leg_seconds["cpu_baseline"] = round(time.perf_counter() - t_leg, 3)
        # where the wall time of this command went: the GPU is idle during the CPU-baseline leg (a sampled gpu_busy of 0 % has its reason here)
        out["leg_seconds"] = leg_seconds
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
