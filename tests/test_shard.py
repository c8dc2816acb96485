"""Multi-GPU path on CPU: row-band partition + the single gather, exercised with
torch.distributed's gloo backend at world_size 2 (the oracle stands in for the renderer —
the collective logic under test is ptmi/shard.py, the same code bench.py runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

from ptmi import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_partition():
    for H, N in ((1080, 1), (1080, 2), (1080, 8), (2160, 8), (17, 4), (8, 8)):
        bands = [shard.band(H, N, r) for r in range(N)]
        assert bands[0][0] == 0 and bands[-1][1] == H
        assert all(bands[i][1] == bands[i + 1][0] for i in range(N - 1))
        sizes = [b - a for a, b in bands]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    with pytest.raises(ValueError):
        shard.band(3, 4, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, frames, q):
    sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_lib import Oracle
    from ptmi import layout, scenes, shard as sh
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.make("cornell")
        cam = layout.make_camera(W, H)
        y0, y1 = sh.band(H, world, rank)
        frame = np.zeros((H, W, 4), np.float32)
        Oracle().render(sc, cam, frames, out=frame, y0=y0, y1=y1, threads=2)      # this rank's rows only
        assert not frame[:y0].any() and not frame[y1:].any()
        t = torch.from_numpy(frame)
        sh.gather_bands(dist, t, H, world, rank)
        dist.barrier()
        if rank == 0:
            full, _ = Oracle().render(sc, cam, frames, threads=2)
            q.put(bool(np.array_equal(t.numpy().view(np.uint32), full.view(np.uint32))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [32, 33])                  # equal bands (zero-copy views) and ragged bands (padded)
def test_two_rank_gloo_gather_equals_single_render(H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, 40, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_weak_frame_keeps_the_view_and_the_work_per_rank():
    for N in (1, 2, 4, 8):
        w, h = shard.weak_frame(1920, 1080, N)
        assert h % (shard.STRIP_ROWS * N) == 0 and w % 16 == 0
        assert abs(w / h - 16 / 9) < 0.01                                  # same aspect -> same picture
        assert abs(w * h / N / (1920 * 1080) - 1) < 0.01                   # same pixels per rank
    assert shard.weak_frame(1920, 1080, 1) == (1920, 1080) and shard.weak_frame(1920, 1080, 4) == (3840, 2160)
    assert shard.strip_options(1, 0)["tile_parts"] == 0
    assert shard.strip_options(8, 5) == dict(tile_y0=0, tile_y1=0, tile_parts=8, tile_part=5, tile_strip=shard.STRIP_ROWS)
    with pytest.raises(ValueError):
        shard.strip_options(4, 4)


def _strip_worker(rank, world, port, H, W, frames, q):
    sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_lib import Oracle
    from ptmi import layout, scenes, shard as sh
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.make("cornell")
        cam = layout.make_camera(W, H)
        frame = np.zeros((H, W, 4), np.float32)
        o, strip = Oracle(), sh.strip_rows_for(H, world)
        for y0 in range(rank * strip, H, world * strip):                          # this rank's strips only
            o.render(sc, cam, frames, out=frame, y0=y0, y1=min(y0 + strip, H), threads=2)
        assert sorted(np.flatnonzero(frame.any(axis=(1, 2)))) == sh.strip_rows(H, world, rank, strip)
        t = torch.from_numpy(frame)
        sh.gather_strips(dist, t, world, rank, strip)
        dist.barrier()
        if rank == 0:
            full, _ = Oracle().render(sc, cam, frames, threads=2)
            q.put(bool(np.array_equal(t.numpy().view(np.uint32), full.view(np.uint32))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [32, 30, 27])      # whole rounds of 4-row strips; 30 % 8 != 0 -> 3-row strips; 27: 1-row strips, odd count
def test_two_rank_gloo_strip_gather_equals_single_render(H):
    """Interleaved strips (the bench's shard unit): rank r renders strips r, r + 2, ...; one gather of packed rows
    de-interleaves them into the frame of a single render. gather_strips is the code bench.py runs over RCCL: it only
    touches contiguous send / receive buffers, so the backend sees the same tensors here and there."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, 2, port, H, 40, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


class _LoopbackDist:
    """Stands in for torch.distributed in one process: gather() records what each rank sends and hands the root the
    list a real gather would deliver. Lets the packing / unpacking run for world sizes the CPU suite does not spawn."""
    def __init__(self, world):
        self.world, self.sent = world, {}

    def run(self, frames, strip, fn):
        for r in range(1, self.world):                       # non-root ranks first: they only send
            self.rank = r
            fn(self, frames[r], self.world, r, strip)
        self.rank = 0
        return fn(self, frames[0], self.world, 0, strip)

    def gather(self, send, recv, dst=0):
        assert send.is_contiguous()
        if self.rank != dst:
            assert recv is None
            self.sent[self.rank] = send.clone()
            return
        assert len(recv) == self.world and all(t.is_contiguous() and t.shape == send.shape for t in recv)
        for r in range(self.world):
            recv[r].copy_(send if r == dst else self.sent[r])


@pytest.mark.parametrize("H,world", [(2160, 8), (64, 8), (37, 3), (5, 4), (1080, 1)])
def test_strip_rows_partition_and_loopback_gather(H, world):
    """configs[4]'s frame over 8 ranks (2160 rows -> 3-row strips), ragged frames and more ranks than strips: the rows of
    all ranks partition the frame, and gather_strips reassembles a frame whose row y holds the value y."""
    import torch
    strip = shard.strip_rows_for(H, world)
    if (H, world) == (2160, 8):
        assert strip == 3                                    # 2160 % (4 * 8) != 0, 2160 % (3 * 8) == 0
    rows = [shard.strip_rows(H, world, r, strip) for r in range(world)]
    assert sorted(sum(rows, [])) == list(range(H))
    if H % (strip * world) == 0:
        assert len({len(r) for r in rows}) == 1              # whole rounds: equal shares
    W = 3
    want = torch.arange(H, dtype=torch.float32).view(H, 1, 1).expand(H, W, 4).contiguous()
    frames = []
    for r in range(world):
        f = torch.full((H, W, 4), -1.0)
        if rows[r]:
            f[rows[r]] = want[rows[r]]
        frames.append(f)
    got = _LoopbackDist(world).run(frames, strip, lambda d, f, w, r, s: shard.gather_strips(d, f, w, r, s))
    assert torch.equal(got, want)


def _strip8_worker(rank, world, port, H, W, q):
    sys.path.insert(0, os.path.join(ROOT, "wgpu-path-tracing_amd"))
    import torch
    import torch.distributed as dist
    from ptmi import shard as sh
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        strip = sh.strip_rows_for(H, world)
        rows = sh.strip_rows(H, world, rank, strip)
        frame = torch.zeros((H, W, 4), dtype=torch.float32)
        g = sh.StripGather(frame, world, rank, strip)          # built once, outside the loop, as bench.py does
        ok = True
        for rnd in (1, 2):                                     # two gathers through the same buffers (warm-up + timed region)
            frame.zero_()
            for y in rows:                                     # a value that names (round, row, column, channel)
                frame[y] = (rnd * 1e6 + y * 16 + torch.arange(W * 4, dtype=torch.float32).reshape(W, 4) % 16)
            g.run(dist)
            dist.barrier()
            if rank == 0:
                want = (rnd * 1e6 + torch.arange(H, dtype=torch.float32).reshape(H, 1, 1) * 16
                        + (torch.arange(W * 4, dtype=torch.float32).reshape(1, W, 4) % 16))
                ok = ok and bool(torch.equal(frame, want))
        if rank == 0:
            q.put((ok, strip, len(rows)))
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_strip_gather_of_the_2160_row_frame():
    """`bench.py --config 4 --gpus 8` end to end on the CPU: 2160 rows are not a whole number of rounds of 8 x 4 rows, so the
    shard unit becomes 3-row strips (shard.strip_rows_for); eight gloo ranks each fill their 270 rows, ONE StripGather per rank
    runs twice, and the root must hold every row of the frame, in place, both times."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip8_worker, args=(r, 8, port, 2160, 6, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, strip, n_rows = q.get(timeout=10)
    assert (ok, strip, n_rows) == (True, 3, 270)
