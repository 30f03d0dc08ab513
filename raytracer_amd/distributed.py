"""Multi-GPU render: interleaved row tiles per rank, one exchange step (gather to rank 0).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank
renders tiles t = rank, rank+G, ... of the frame with the same keyed RNG, so the assembled image
is byte-identical to a 1-GPU render.  The only collective is a fixed-size
``gather`` of the packed uint8 slabs to rank 0 — G-1 concurrent point-to-point transfers, one per
xGMI link into GPU 0 — followed by a row scatter on the root (rt_scatter_rows).  The reference has
no distributed code; its unused ``render_chunk`` tile split (raytracer.py:299-308) is the model.
"""
from __future__ import annotations

import numpy as np

from .tiling import DEFAULT_TILE_ROWS, local_rows, max_local_rows


def gather_row_tiles(local, width: int, height: int, tile_rows: int, group=None, renderer=None,
                     dst: int = 0, slab=None, gathered=None, out=None):
    """Gather every rank's packed rows to ``dst`` and put them in image order.

    ``local``: uint8 tensor [rows_of_this_rank, W, 3] as packed by ``Renderer.render_rows``.
    Works on GPU tensors (nccl; root placement by the HIP scatter kernel) and on CPU tensors
    (gloo; root placement by index_copy_ — used by the CPU tests of this exchange step, no pixel
    is computed here).  Returns the image [H, W, 3] on ``dst`` and None elsewhere.
    ``slab`` / ``gathered`` / ``out`` let callers reuse buffers across frames.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    rows_max = max_local_rows(height, tile_rows, world)
    stride = rows_max * width * 3
    if slab is None:
        slab = torch.empty(stride, dtype=torch.uint8, device=local.device)
    n = local.numel()
    if n and local.data_ptr() != slab.data_ptr():      # render_rows may have written in place
        slab[:n].copy_(local.reshape(-1))
    if world == 1:
        parts = None
        gathered = slab.view(1, stride)
    else:
        if rank == dst:
            if gathered is None:
                gathered = torch.empty((world, stride), dtype=torch.uint8, device=local.device)
            parts = list(gathered.unbind(0))
        else:
            parts = None
        dist.gather(slab, parts, dst=dst, group=group)
    if rank != dst:
        return None
    if local.is_cuda:
        if renderer is None:
            raise ValueError("a Renderer is required to place GPU tiles (rt_scatter_rows)")
        return renderer.scatter_rows(gathered, world, width, height, tile_rows, out=out)
    if out is None:
        out = torch.empty((height, width, 3), dtype=torch.uint8)
    for r in range(world):
        rows = torch.from_numpy(local_rows(height, tile_rows, r, world))
        out.index_copy_(0, rows, gathered[r, :len(rows) * width * 3].view(len(rows), width, 3))
    return out


class BatchExchange:
    """Double-buffered exchange of frame batches: the gather of batch i runs (on the collective's own
    stream) while batch i + 1 is rendered, and the root places batch i after it has launched batch i + 1.

    Per step every rank calls ``slab = ex.slab(i)``, renders its row tiles of the batch into it, then
    ``ex.submit(i)``.  ``ex.drain()`` finishes what is in flight.  ``place(gathered, i)`` runs on the root
    when batch i has arrived (``gathered``: [world, F, slab_bytes]).  Buffers: two slabs per rank, two
    gather buffers on the root.  With ``world == 1`` the slab is handed to ``place`` directly.
    Works with nccl (GPU tensors) and gloo (CPU tensors; used by the CPU tests of this logic).

    ``sparse=True`` sends only what is not black.  The exchange, not the rays, bounds N > 1 (one GPU renders
    several times what an xGMI link carries), and most of a frame is sky: every rank keeps the
    32-byte segments of its slab that hold a non-zero byte (19 % of config 2), the ranks agree on the largest
    count (one tiny all_gather), and the same fixed-size ``gather`` moves [count_max, 5] int64 rows — segment
    number + 32 bytes — instead of the slab; the root zero-fills its gather buffer and puts the segments back
    before ``place`` runs.  Lossless: ``place`` sees the same bytes either way (tests).  Needs the batch's
    bytes to be a multiple of 32 (else the dense path is used)."""

    def __init__(self, n_frames: int, slab_bytes: int, device, place, group=None, dst: int = 0,
                 sparse: bool = False, place_segments=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.dst, self.place = dist, group, dst, place
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.slabs = [torch.zeros((n_frames, slab_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.sparse = bool(sparse) and self.world > 1 and (n_frames * slab_bytes) % 32 == 0
        # sparse with ``place_segments(rows, counts, step)``: the root hands the gathered [world, count_max, 5]
        # segment rows to the caller as they are (see segments_to_frames) and keeps no dense gather buffer
        self.place_segments = place_segments if self.sparse else None
        self.gathered = [torch.empty((self.world, n_frames, slab_bytes), dtype=torch.uint8, device=device)
                         for _ in range(2)] if (self.rank == dst and self.world > 1 and self.place_segments is None) \
            else [None, None]
        self.pending = [None, None]      # (work handle, step[, segments, counts]) of the batch in flight in each buffer
        self.sent_bytes = 0              # bytes this rank handed to the gather so far (either path)

    def slab(self, step: int):
        self._finish(step & 1)           # the batch that used this buffer two steps ago must be gone
        return self.slabs[step & 1]

    def submit(self, step: int):
        b = step & 1
        if self.world == 1:
            self.place(self.slabs[b].unsqueeze(0), step)
            return
        if self.sparse:
            self._submit_sparse(step)
            return
        self.sent_bytes += self.slabs[b].numel()
        if self.slabs[b].is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal of the N > 1 path on one GPU (bench.py RT_BENCH_BACKEND=gloo): gloo has no gather for
            # device tensors, so the slabs go through host memory, synchronously
            import torch
            host = self.slabs[b].cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            self.dist.gather(host, parts, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                self.gathered[b].copy_(torch.stack(parts))
            self.pending[b] = (None, step)
            self._finish(1 - b)
            return
        parts = list(self.gathered[b].unbind(0)) if self.rank == self.dst else None
        work = self.dist.gather(self.slabs[b], parts, dst=self.dst, group=self.group, async_op=True)
        self.pending[b] = (work, step)
        self._finish(1 - b)              # the previous batch: its gather overlapped this batch's render

    def _submit_sparse(self, step: int):
        import torch
        b = step & 1
        dist, slab = self.dist, self.slabs[b]
        staged = slab.is_cuda and dist.get_backend(self.group) == "gloo"   # one-GPU rehearsal: through host memory
        seg = slab.view(-1).view(torch.int64).view(-1, 4)                  # 32-byte segments
        nz = (seg != 0).any(1).nonzero().squeeze(1)                        # (waits for the batch to be rendered)
        n = int(nz.numel())
        cdev = "cpu" if staged else slab.device
        mine = torch.tensor([n], dtype=torch.int64, device=cdev)
        every = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(self.world)]
        dist.all_gather(every, mine, group=self.group)
        counts = [int(c) for c in torch.cat(every).tolist()]
        nmax = max(counts)
        self.sent_bytes += nmax * 40
        work = rows = None
        if nmax:
            send = torch.empty((nmax, 5), dtype=torch.int64, device=slab.device)
            send[:n, 0] = nz
            send[:n, 1:] = seg.index_select(0, nz)
            if n < nmax:
                send[n:].zero_()
            if staged:
                host = send.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
                dist.gather(host, parts, dst=self.dst, group=self.group)
                if self.rank == self.dst:
                    rows = torch.stack(parts).to(slab.device)
            else:
                if self.rank == self.dst:
                    rows = torch.empty((self.world, nmax, 5), dtype=torch.int64, device=slab.device)
                work = dist.gather(send, list(rows.unbind(0)) if self.rank == self.dst else None, dst=self.dst,
                                   group=self.group, async_op=True)
        self.pending[b] = (work, step, rows, counts, send if nmax else None)
        self._finish(1 - b)

    def _finish(self, b: int):
        if self.pending[b] is None:
            return
        work, step = self.pending[b][:2]
        extra = self.pending[b][2:]
        self.pending[b] = None
        if work is not None:
            work.wait()                  # nccl: the current stream waits; gloo: the host does
        if self.rank == self.dst:
            if extra and self.place_segments is not None:
                self.place_segments(extra[0], extra[1], step)
                return
            if extra:                    # sparse: zero-fill, then put every rank's segments back
                import torch
                rows, counts = extra[0], extra[1]
                dense = self.gathered[b].view(self.world, -1).view(torch.int64).view(self.world, -1, 4)
                dense.zero_()
                for r, k in enumerate(counts):
                    if k:
                        dense[r].index_copy_(0, rows[r, :k, 0], rows[r, :k, 1:])
            self.place(self.gathered[b], step)

    def drain(self):
        # in submission order
        order = sorted((p[1], b) for b, p in enumerate(self.pending) if p is not None)
        for _, b in order:
            self._finish(b)


def segments_to_frames(rows, counts, frames_out, width: int, height: int, tile_rows: int, rows_max: int):
    """Root side of the sparse exchange without a dense intermediate: ``rows`` [world, count_max, 5] int64 (segment
    number within the sender's [F, rows_max, W * 3] slab + its 32 bytes; None when every count is 0) go straight
    to their place in ``frames_out`` uint8 [F, H, W, 3], which is zero-filled first.  Needs W * 3 % 32 == 0."""
    import torch
    world = len(counts)
    spr = width * 3 // 32                       # segments per image row
    assert width * 3 % 32 == 0 and frames_out.is_contiguous()
    out = frames_out.view(-1).view(torch.int64).view(-1, 4)
    out.zero_()
    for r, k in enumerate(counts):
        if not k:
            continue
        rowmap = torch.from_numpy(local_rows(height, tile_rows, r, world).astype(np.int64)).to(rows.device)
        s = rows[r, :k, 0]
        f = torch.div(s, rows_max * spr, rounding_mode="floor")
        rem = s - f * (rows_max * spr)
        lrow = torch.div(rem, spr, rounding_mode="floor")
        dest = (f * height + rowmap[lrow]) * spr + (rem - lrow * spr)
        out.index_copy_(0, dest, rows[r, :k, 1:])
    return frames_out


def render_distributed(renderer, width, height, samples=1, *, tile_rows=DEFAULT_TILE_ROWS,
                       group=None, max_depth=5, camera=None, rng_mode="keyed", seed=0,
                       buffers: dict | None = None):
    """Render one frame across all ranks of ``group``; rank 0 gets the uint8 [H, W, 3] tensor.

    The scene must already be uploaded on every rank (``renderer.upload_scene``)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b = buffers if buffers is not None else {}
    local = renderer.render_rows(width, height, samples, tile_rows=tile_rows, tile_first=rank,
                                 tile_stride=world, max_depth=max_depth, camera=camera,
                                 rng_mode=rng_mode, seed=seed, out=b.get("local"))
    return gather_row_tiles(local, width, height, tile_rows, group=group, renderer=renderer,
                            slab=b.get("slab"), gathered=b.get("gathered"), out=b.get("out"))


def make_buffers(renderer, width, height, tile_rows, world, rank):
    """Pre-allocate the per-frame buffers used by render_distributed (animation / bench loops)."""
    import torch
    dev = f"cuda:{renderer.device}"
    rows_max = max_local_rows(height, tile_rows, world)
    slab = torch.empty(rows_max * width * 3, dtype=torch.uint8, device=dev)
    b = dict(slab=slab, local=slab.view(rows_max, width, 3))   # rows are rendered into the slab
    if rank == 0:
        b["gathered"] = torch.empty((world, rows_max * width * 3), dtype=torch.uint8, device=dev)
        b["out"] = torch.empty((height, width, 3), dtype=torch.uint8, device=dev)
    return b


def frames_of_rank(n_frames: int, rank: int, world: int) -> np.ndarray:
    """Frame-parallel animation (config 5): frame f -> rank f mod G."""
    return np.arange(rank, n_frames, world)
