#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X wavefront path tracer.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic Cornell box, 1920x1080, 8 bounces, MIS on.
One step = one ptmi_dispatch of --frames-per-step frames (default 32, traced as one wavefront
batch of 66 M paths) over the rank's rows; the default K = 2 steps therefore render exactly the
64 spp of configs[1]. Scene and output
live in HBM before the timed region starts (the C ABI copies host blobs at upload).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling over pixel rows —
the same view at N times the pixels (1920x1080, 2720x1528, 3840x2160, 5424x3040: same aspect,
so the same mix of cheap and expensive pixels per rank), rank r renders the 4-row strips
r, r + N, ... with no data-path collective (pixels and RNG streams are independent,
pt.wgsl:719, :753-761); after the last step the strips are gathered to rank 0 with ONE RCCL
gather, inside the timed region. value = all ranks' path segments / max-over-ranks time.

Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (closest-hit
traversal `extend`) against HBM: algorithmic bytes per ray = 32 (origin+direction
float4 pair) + 4 (queue index) + 16 (hit record) = 52 B, launch durations
from HIP events recorded by the library on its own stream around every extend launch.
`cpu_baseline` times the CPU oracle (oracle/, a restatement — the reference has no CPU
path) on a bounded crop of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

EXTEND_BYTES_PER_RAY = 32 + 4 + 16
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def pipeline_bytes_per_segment(do_mis, mean_len):
    """SURVEY.md §8(d) re-derived for this build's records:
    extend R 32+4 W 16; shade R 4+16+48 (O,D,T) W 48 + masks 0.25; compact R 4 W 4;
    MIS: shadow record W 48 R 48 + radiance RMW 32; per path: raygen W 48 (O, D, L), accumulate R 16 + frame RMW 32,
    less what bounce 0 does not touch (the identity queue, 3 x 4, and the stored throughput, 16)."""
    seg = (32 + 4 + 16) + (4 + 16 + 48 + 48) + 8
    if do_mis:
        seg += 48 + 48 + 32
    return seg + (48 + 48 - 12 - 16) / max(mean_len, 1e-9)


def pmc_traffic(default_workload):
    """HBM bytes per extend launch from the committed rocprofv3 PMC passes of this exact command
    (profiles/r01_bench_n1_pmc.json: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, KB per
    launch; gfx950 tallies wide reads at half: traffic = 2*FETCH_SIZE + WRITE_SIZE). None when the
    workload differs from the profiled one — counters cannot be read from inside the timed run."""
    path = os.path.join(ROOT, "profiles", "r01_bench_n1_pmc.json")
    if not default_workload or not os.path.exists(path):
        return None
    d = json.load(open(path))
    try:
        f = d["FETCH_SIZE"]["k_trace_lds/extend"]["avg_KB_per_launch"]
        w = d["WRITE_SIZE"]["k_trace_lds/extend"]["avg_KB_per_launch"]
    except KeyError:
        return None
    return int((2 * f + w) * 1024)


def valu_issue(avg_launch_ms, default_workload):
    """SIMD cycles per VALU wave-instruction of the extend kernel: instructions per launch from the committed
    rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/r01_bench_n1_valu.json), duration measured live."""
    path = os.path.join(ROOT, "profiles", "r01_bench_n1_valu.json")
    if not default_workload or not os.path.exists(path) or not avg_launch_ms:
        return None
    try:
        n = json.load(open(path))["SQ_INSTS_VALU"]["k_trace_lds/extend"]["avg_per_launch"]
    except KeyError:
        return None
    return {"wave_instructions_per_launch": int(n),
            "simd_cycles_per_instruction": round(avg_launch_ms * 1e-3 * 2.4e9 * 1024 / n, 3),
            "microbenchmark_cycles_per_instruction": 3.0}


def cpu_baseline(scene, width, height, bounces, mis, threads, target_s=12.0):
    """Oracle on whole frames of the same camera: 1 calibration frame, then as many frames as
    fit in about target_s seconds (at most the 64 of the workload)."""
    from oracle_lib import Oracle
    from ptmi import layout
    import numpy as np
    orc = Oracle(strict=False)
    out = np.zeros((height, width, 4), np.float32)
    _, st0 = orc.render(scene, layout.make_camera(width, height), 1, max_bounces=bounces, do_mis=mis, out=out,
                        threads=threads)
    frames = int(max(1, min(63, target_s / max(st0.seconds, 1e-3))))
    _, st = orc.render(scene, layout.make_camera(width, height, frame_index=1), frames, max_bounces=bounces,
                       do_mis=mis, out=out, threads=threads)
    return {
        "value": round(st.segments / st.seconds / 1e6, 4), "unit": "Msamples/s", "cores": int(st.threads),
        "kind": "port",
        "sample": f"frames 1..{frames} of the full {width}x{height} frame: {st.paths} paths, {st.segments} segments "
                  f"in {st.seconds:.2f} s (oracle/pt_oracle.c contract build, OpenMP dynamic rows)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=32)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080, help="rows per GPU")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--no-mis", action="store_true")
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--frames-per-batch", type=int, default=0)
    ap.add_argument("--traversal", default="auto", choices=["auto", "global", "lds"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keep-reference-tree", action="store_true",
                    help="walk the BVH exactly as uploaded instead of the hierarchy rebuilt over its leaves")
    ap.add_argument("--rehearse", action="store_true",
                    help="dry run of the N>1 path on ONE GPU: every rank uses device 0, gloo backend, bands gathered "
                         "through host memory, rank 0 checks the gathered frame bit for bit against its own unsharded "
                         "render. Not a benchmark.")
    ap.add_argument("--timing", type=int, default=2, help="library HIP-event timing level (2 = every extend launch, 3 = every kernel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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

    mis = 0 if args.no_mis else 1
    # Weak scaling: the SAME view at `world` times the pixels (1920x1080 -> 2720x1528 -> 3840x2160 -> 5424x3040), so
    # the mix of cheap and expensive pixels does not change with N; rank r renders the 4-row strips r, r + N, ...
    # (an even sample of the picture: no rank is stuck with the expensive rows), gathered once at the end.
    W, H = shard.weak_frame(args.width, args.height, world)
    scene = scenes.make(args.scene)

    ctx = native.Context(local_rank)
    ctx.set_options(keep_reference_tree=int(args.keep_reference_tree))
    ctx.upload_scene(scene)
    ctx.resize(W, H)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")       # binding 0, owned by the caller
    ctx.bind_output_device(frame.data_ptr(), frame.numel() * 4)
    # One explicit stream for rendering, copies and the collective. (torch's default stream has the NULL
    # handle, which ptmi_set_stream reads as "use the context's own stream"; that stream is not ordered
    # against torch work, so a gather could overtake the render.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    trav = {"auto": native.TRAVERSAL_AUTO, "global": native.TRAVERSAL_GLOBAL, "lds": native.TRAVERSAL_LDS}[args.traversal]
    ctx.set_options(max_bounces=args.bounces, do_mis=mis, frames_per_batch=args.frames_per_batch, traversal=trav, cull=1,
                    timing=args.timing, **shard.strip_options(world, rank))

    fps = args.frames_per_step
    frame_index = 0

    def step():
        nonlocal frame_index
        ctx.dispatch(layout.make_camera(W, H, frame_index=frame_index), fps)
        frame_index += fps

    def gather():
        # the bands accumulate locally; ONE gather assembles the frame after the last frame
        if world > 1 and args.rehearse:
            host = frame.cpu()
            shard.gather_strips(dist, host, world, rank)
            if rank == 0:
                frame.copy_(host)
        elif world > 1:
            shard.gather_strips(dist, frame, world, rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gather()                                  # also sets up the RCCL channels outside the timed region
    fence()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gather()
    fence()
    dt = time.perf_counter() - t0

    st = ctx.stats()
    rehearsal_ok = None
    if args.rehearse and rank == 0:
        # the same frames, unsharded, on this rank alone: the sharded + gathered frame must equal it bit for bit
        gathered = frame.cpu().numpy().copy()
        ctx.set_options(tile_y0=0, tile_y1=0, tile_parts=0, tile_part=0, timing=0)
        frame.zero_()
        for k in range(args.warmup + args.steps):
            ctx.dispatch(layout.make_camera(W, H, frame_index=k * fps), fps)
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
        ext_ms = st.extend_ms / max(st.extend_launches, 1)
        ext_gbs = (st.segments * EXTEND_BYTES_PER_RAY / 1e9) / (st.extend_ms / 1e3) if st.extend_ms > 0 else None
        b_seg = pipeline_bytes_per_segment(mis, mean_len)
        out = {
            "metric": "Msamples/s (rays x bounces / s)", "value": round(msamples, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.scene} {args.width}x{args.height} pixels per GPU, {args.steps * fps} spp, {args.bounces} bounces, "
                            f"MIS {'on' if mis else 'off'} (BASELINE.json configs[1]); frame {W}x{H}",
                "frames_per_step": fps, "frames_per_batch": int(st.frames_per_batch_used),
                "traversal": "lds" if st.traversal_used == native.TRAVERSAL_LDS else "global",
                "triangles": int(len(scene.tris)), "bvh_nodes": int(len(scene.nodes)), "parallelism": f"{shard.STRIP_ROWS}-row strips x{world}",
            },
            **({"rehearsal": {"sharded_equals_unsharded_bitwise": rehearsal_ok, "backend": "gloo", "note": "all ranks on one GPU; not a benchmark"}} if args.rehearse else {}),
            "segments": int(segments), "shadow_rays": int(shadow_rays), "paths": int(paths),
            "mean_path_length": round(mean_len, 4),
            "nominal_msamples": round(paths * args.bounces / dt / 1e6, 3),
            "gpu_ms_rank0": round(st.gpu_ms, 3),
            "kernel_ms_rank0": {"extend": round(st.extend_ms, 3), "shade": round(st.shade_ms, 3),
                                "shadow": round(st.shadow_ms, 3)},
            "roofline": {
                "bound": "hbm", "kernel": "extend (closest-hit BVH traversal)",
                "achieved": None if ext_gbs is None else round(ext_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None if ext_gbs is None else round(ext_gbs / HBM_PEAK_GBS, 6),
                "traffic": pmc_traffic(args.scene == "cornell" and W == 1920 and args.height == 1080 and fps == 32
                                       and world == 1 and args.traversal == "auto" and mis and args.bounces == 8),
                "traffic_note": "bytes per extend launch, 2*FETCH_SIZE+WRITE_SIZE from profiles/r01_bench_n1_pmc.json",
                "algorithmic_bytes_per_launch": int(st.segments * EXTEND_BYTES_PER_RAY / max(st.extend_launches, 1)),
                "bytes_per_unit": EXTEND_BYTES_PER_RAY, "units_per_launch": round(st.segments / max(st.extend_launches, 1), 1),
                "avg_launch_ms": round(ext_ms, 4), "launches": int(st.extend_launches),
                "pipeline_bytes_per_segment": round(b_seg, 1),
                "pipeline_achieved": round(msamples * 1e6 * b_seg / 1e9, 3),
                "pipeline_frac": round(msamples * 1e6 * b_seg / 1e9 / HBM_PEAK_GBS, 6),
                # why frac is low: the kernel saturates the vector ALUs, not HBM (scene in LDS / L2). VALU instructions per
                # launch from the committed SQ_INSTS_VALU pass of this command; cycles at the nominal 2.4 GHz over 1024 SIMDs
                # (a VALU microbenchmark, tools/ubench/pk.hip, issues one per 3.0 cycles per SIMD at this occupancy)
                "valu": valu_issue(ext_ms, args.scene == "cornell" and W == 1920 and args.height == 1080 and fps == 32
                                   and world == 1 and args.traversal == "auto" and mis and args.bounces == 8),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, args.bounces, mis, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
