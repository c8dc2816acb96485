"""Pixel-row sharding across the GPUs of one node.

The reference renders on one device; pixels are independent (one thread per pixel,
src/shader/pt.wgsl:753-761) and the RNG is seeded per (x, y, frame)
(src/shader/random.wgsl:3-5), so a frame split into contiguous row bands is
bit-identical to the unsplit frame. Each rank renders its band with tile_y0/tile_y1
(include/ptmi.h) — no collective on the data path — and the accumulated bands are
concatenated on rank 0 with ONE gather (RCCL over xGMI on GPUs, gloo in CPU tests).
"""


def band(height, world, rank):
    """Rows [y0, y1) of rank `rank`: contiguous, in rank order, sizes differ by at most one."""
    if not (0 <= rank < world) or height < world:
        raise ValueError(f"cannot split {height} rows over {world} ranks (rank {rank})")
    return rank * height // world, (rank + 1) * height // world


def gather_bands(dist, frame, height, world, rank, dst=0):
    """frame: (height, width, 4) tensor, on every rank holding valid rows only in that rank's band.
    After the call rank `dst` holds the complete frame. Works with any torch.distributed backend."""
    if world == 1:
        return frame
    bands = [band(height, world, r) for r in range(world)]
    sizes = {y1 - y0 for y0, y1 in bands}
    y0, y1 = bands[rank]
    if len(sizes) == 1:
        # equal bands: gather straight into row views of the destination frame (concatenation)
        gl = [frame[a:b] for a, b in bands] if rank == dst else None
        # the destination's own band must not alias its slot in the gather list
        mine = frame[y0:y1].clone() if rank == dst else frame[y0:y1]
        dist.gather(mine, gl, dst=dst)
        return frame
    hmax = max(sizes)
    import torch
    send = torch.zeros((hmax,) + tuple(frame.shape[1:]), dtype=frame.dtype, device=frame.device)
    send[: y1 - y0] = frame[y0:y1]
    gl = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gl, dst=dst)
    if rank == dst:
        for r, (a, b) in enumerate(bands):
            if r != dst:
                frame[a:b] = gl[r][: b - a]
    return frame


# ---- interleaved strips ("if imbalance shows, switch to 16-row strips assigned round-robin and
# de-interleave after the gather") -----------------------------------------------------------------------------
STRIP_ROWS = 4


def weak_frame(base_width, base_height, world, strip=STRIP_ROWS):
    """Frame of `world` times the pixels of base_width x base_height with the SAME aspect (so the same view and the
    same mix of cheap and expensive pixels at every world size), its height a whole number of strip rounds."""
    if world == 1:
        return base_width, base_height
    s = world ** 0.5
    rnd = strip * world
    h = max(rnd, int(round(base_height * s / rnd)) * rnd)
    w = max(16, int(round(base_width * s / 16)) * 16)
    return w, h


def strip_options(world, rank, strip=STRIP_ROWS):
    """ptmi options (include/ptmi.h) that make rank `rank` render strips rank, rank + world, ... of the frame."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return dict(tile_y0=0, tile_y1=0, tile_parts=world if world > 1 else 0, tile_part=rank if world > 1 else 0,
                tile_strip=strip)


def strip_rows_for(height, world, strip=STRIP_ROWS):
    """Strip height for `world` ranks: STRIP_ROWS when the frame is a whole number of rounds of that, else the largest
    smaller height that is (3840x2160 over 8 ranks: 2160 % 32 != 0 -> 3-row strips). Ragged frames work with any strip
    (the last round is short); a whole number of rounds only keeps every rank's share equal."""
    if world <= 1:
        return strip
    for s in range(strip, 0, -1):
        if height % (s * world) == 0:
            return s
    return strip


def strip_rows(height, world, rank, strip=STRIP_ROWS):
    """Frame rows of rank `rank`, ascending: strips rank, rank + world, ... of `strip` rows (the last may be short) —
    the rows include/ptmi.h's tile_parts / tile_part / tile_strip make a context render (DevBand::row_of)."""
    rows = []
    for s0 in range(rank * strip, height, world * strip):
        rows.extend(range(s0, min(s0 + strip, height)))
    return rows


class StripGather:
    """Everything the gather of interleaved strips needs, built ONCE: the row indices of every rank (on the frame's device), the
    packed send buffer and the root's receive buffers. run() then only packs (index_select into the send buffer), gathers ONE
    list of equal buffers and unpacks by row index on the root — no allocation, no host-to-device copy inside a timed region.
    The same code runs on CPU tensors with gloo and on device tensors with RCCL."""

    def __init__(self, frame, world, rank, strip=STRIP_ROWS, dst=0):
        import torch
        self.frame, self.world, self.rank, self.strip, self.dst = frame, world, rank, strip, dst
        h = frame.shape[0]
        self.rows = [strip_rows(h, world, r, strip) for r in range(world)]
        self.nmax = max(len(r) for r in self.rows) if world > 1 else 0
        if world == 1:
            return
        self.idx = [torch.tensor(r, dtype=torch.long, device=frame.device) for r in self.rows]
        self.send = frame.new_zeros((self.nmax,) + tuple(frame.shape[1:]))
        self.recv = [torch.empty_like(self.send) for _ in range(world)] if rank == dst else None

    def run(self, dist):
        if self.world == 1:
            return self.frame
        import torch
        n = len(self.rows[self.rank])
        if n:
            torch.index_select(self.frame, 0, self.idx[self.rank], out=self.send[:n])
        dist.gather(self.send, self.recv, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst and len(self.rows[r]):
                    self.frame.index_copy_(0, self.idx[r], self.recv[r][: len(self.rows[r])])
        return self.frame


def gather_strips(dist, frame, world, rank, strip=STRIP_ROWS, dst=0):
    """frame: (height, width, 4) tensor whose rows strip_rows(height, world, rank, strip) are valid on rank `rank`.
    ONE gather of contiguous buffers (each rank packs its rows; the root unpacks them by row index); afterwards rank
    `dst` holds the complete frame. A loop that gathers repeatedly builds a StripGather once and calls run()."""
    return StripGather(frame, world, rank, strip, dst).run(dist)
