"""Dry run of the N > 1 path of bench.py on ONE GPU: two ranks (both on device 0), gloo rendezvous, each rank
renders its row band through the C ABI, the bands are gathered, and rank 0 checks the gathered frame bit for
bit against its own unsharded render of the same frames. (RCCL itself needs one GPU per rank and is exercised
by the driver's multi-GPU run; the partition / ordering / gather logic is what this covers.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_one_gpu_sharded_equals_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--width", "640",
           "--height", "180", "--frames-per-step", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "4-row strips x2"
    assert d["rehearsal"]["sharded_equals_unsharded_bitwise"] is True
    from ptmi import shard
    w, h = shard.weak_frame(640, 180, 2)                      # same view, twice the pixels
    assert d["paths"] == w * h * 8 and d["value"] > 0


@pytest.mark.gpu
def test_three_ranks_one_gpu_strong_scaling_of_a_ragged_frame():
    """bench.py --config 4 (strong scaling: the frame is fixed and split) with three ranks on one GPU and a frame height that
    is not a whole number of strip rounds for three ranks (110 rows in 4-row strips: shares of 38, 36 and 36 rows; the
    depth-of-field camera of configs[4]): packed-row gather of unequal shares, rank 0 compares with its own unsharded render."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--config", "4", "--gpus", "3", "--rehearse", "--width", "320",
           "--height", "110", "--frames-per-step", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["config_index"] == 4
    assert d["rehearsal"]["sharded_equals_unsharded_bitwise"] is True
    assert d["paths"] == 320 * 110 * 6                        # the frame itself, not three times it


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """bench.py --gpus 2 outside any launcher: it spawns `python -m torch.distributed.run` itself (fresh child processes, before
    anything has touched the GPU) and passes rank 0's JSON line through — the command the driver's multi-GPU run issues needs no wrapper."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--width", "320", "--height", "96",
           "--frames-per-step", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rehearsal"]["sharded_equals_unsharded_bitwise"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("config,n", [(1, 3), (4, 2)])
def test_bench_single_process_drives_the_multi_handle(config, n):
    """bench.py --single-process: ONE process, ptmi_multi_* (the reference's own model: one host thread, one Renderer —
    src/renderer/renderer.ts:415-454) with the library's pack -> gather -> unpack in the timed region. Rehearsed on one GPU with
    loopback copies in place of the collective; the gathered frame equals one context's unsharded render."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--gpus", str(n), "--single-process", "--rehearse",
           "--width", "320", "--height", "110", "--frames-per-step", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == n and d["rehearsal"]["sharded_equals_unsharded_bitwise"] is True
    assert d["enqueue_ms_per_step"]["mean"] > 0 and d["gather_ms"] >= 0 and d["value"] > 0
