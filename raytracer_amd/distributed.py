"""Multi-GPU render: interleaved row tiles per rank, one exchange step (gather to rank 0); frame-parallel
animation batches (config 5).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank renders tiles
t = rank, rank+G, ... of the frame with the same keyed RNG, so the assembled image is byte-identical to a
1-GPU render (refractive scenes included: their pixel sums are integers).  The only exchange on the
data path is a ``gather`` to rank 0 — G-1 concurrent point-to-point transfers, one per xGMI link into GPU 0 —
followed by the placement on the root.  With ``BatchExchange(comm=renderer)`` that gather and the count exchange
before it run INSIDE librt355 (rt_comm_*: the library's own RCCL communicator, grouped ncclSend / ncclRecv of
exactly the bytes each rank has); without it they go through torch.distributed's collectives.  The reference has no distributed code; its
unused ``render_chunk`` tile split (raytracer.py:299-308) is the model.  Development boxes have one GPU: the RCCL
path has run there as a group of ONE rank (``force_collectives``: every collective call of the N > 1 path, device
tensors, the side stream — tests/test_gpu_rccl.py, bench.py with RT_BENCH_FORCE_EXCHANGE=1), several ranks only over
gloo.
"""
from __future__ import annotations

import os
import numpy as np

from .tiling import DEFAULT_TILE_ROWS, local_rows, max_local_rows


def _staged(t, dist, group) -> bool:
    """Device tensors under gloo (the one-GPU rehearsal of the N > 1 path): collectives go through host memory."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


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
    """Pipelined exchange of frame batches: while batch i + 1 is rendered, batch i travels (on the collective's
    own stream), and the root places a batch after it has launched later ones.

    Per step every rank calls ``slab = ex.slab(i)``, renders its row tiles of the batch into it, then
    ``ex.submit(i)``; ``ex.drain()`` finishes what is in flight.  Works with nccl (GPU tensors) and gloo (CPU
    tensors: the CPU tests of this logic; GPU tensors are staged through host memory: one-GPU rehearsal).

    Dense (default): one fixed-size ``gather`` of the slabs; ``place(gathered, i)`` runs on the root when batch i
    has arrived (``gathered``: [world, F, slab_bytes]).  With ``world == 1`` the slab goes to ``place`` directly.

    Sparse (``compact`` and ``place_tiles`` given — bench.py's N > 1 default): only occupied 8x8 tiles travel.  The
    exchange, not the rays, bounds N > 1 (one GPU renders several times what an xGMI link carries),
    and most of a frame is background.  ``compact(slab, entries, count)`` (Renderer.compact_tiles: a HIP kernel
    driven by the occupancy bits the render left behind) packs them into ``entries`` [capacity, entry_bytes] and
    leaves their number on the device — nothing is read back, so a rank launches render i + 1 without waiting for
    render i.  Three stages, one per submit:
      submit(i):   compact batch i; all ranks exchange their counts (a tiny all_gather, asynchronous);
      submit(i+1): the counts of batch i are read (render i + 1 is queued already, so the GPU stays busy while the
                   host waits) and ONE gather moves [max count, entry_bytes] per rank — instead of the slab;
      submit(i+2): ``place_tiles(entries [world, max count, entry_bytes], counts [world] (device), i)`` on the root
                   (Renderer.place_tiles: one pass over the frames, black where no tile lands).
    Buffers: two slabs and three entry buffers per rank (a batch's entries live until its gather is done).

    With RCCL (device tensors, backend nccl) nothing of the exchange is queued on the render stream: the counts of a
    batch are copied to pinned host memory on a SIDE stream that waits for the count exchange alone, and the host reads
    them after an event there — not after the renders it has queued since —; the gather is issued from the side stream
    (RCCL's stream then waits for that batch's compaction, not for later renders); the root's placement runs there too, or —
    where the root renders itself — on the render stream once the gather has landed (see ``place_beside``;
    ``place_tiles`` is called with the stream it must launch on current).  The
    render stream only waits, three batches later, for the gather that read the entry buffer it is about to overwrite.
    ``force_collectives``: a group of one rank still goes through every collective (the one-GPU rehearsal of this path).
    ``contributes`` False (sparse mode): this rank rendered nothing — a root that only gathers and places (bench.py from three
    GPUs on) —: it joins every collective with a count of zero and ``compact`` is not called.
    ``place_beside``: True = the root's placement runs on the side stream, False = on the render stream once the gather has
    landed; default: beside only where the root does not render (``contributes`` False).
    ``side_stream``: the stream for the collectives (default: a high-priority stream of this exchange's own).
    ``defer_place`` (sparse mode): the pipeline gathers but does not place; ``place_pending()`` places what has been gathered
    so far on the stream current at that call (``drain()`` does too).
    The render stream (torch's current stream when ``submit`` is called) should be a stream of its own, NOT HIP's legacy
    default stream: that one synchronises implicitly with every blocking stream of the process, and made to wait for an event
    behind the library's ncclSend / ncclRecv it deadlocked (bench.py: ``with torch.cuda.stream(torch.cuda.Stream())`` and
    ``renderer.use_torch_stream()``).
    ``comm``: a Renderer that holds a communicator (Renderer.comm_init): the two collectives of the sparse exchange are
    then the library's (rt_comm_all_gather_counts, rt_comm_gather — every rank sends exactly its own entries, nothing is
    padded to the largest rank), issued on the side stream; torch.distributed makes no call on the data path.
    """

    def __init__(self, n_frames: int, slab_bytes: int, device, place, group=None, dst: int = 0, *,
                 compact=None, place_tiles=None, capacity: int | None = None, entry_bytes: int = 196,
                 force_collectives: bool = False, comm=None, contributes: bool = True, place_beside: bool | None = None,
                 defer_place: bool = False, side_stream=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.dst, self.place = dist, group, dst, place
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.comm = comm
        self.contributes = bool(contributes)
        # where the root's placement runs (RCCL path): beside the renders, on the side stream — or, by default where the root
        # renders too, on the render stream behind the gather: measured, a placement that competes with the root's own shading
        # kernel costs more than it hides (tools/root_emulation.py: config 4 at N = 2 6.1 ms per batch beside, 5.2 behind)
        self.place_beside = (not self.contributes) if place_beside is None else bool(place_beside)
        # defer_place (sparse mode): a batch whose gather has been issued is not placed by the pipeline but kept until the
        # caller says place_pending() — on the stream that is current then.  For a root that renders in some batches only and
        # wants to place in the others, when its render stream has nothing else to do.
        self.defer_place = bool(defer_place)
        self.deferred = []               # (gathered, counts, step, event after the gather) of batches waiting for place_pending()
        if comm is not None:
            info = comm.comm_info()
            if compact is None or torch.device(device).type != "cuda":
                raise ValueError("comm: the library's collectives serve the sparse exchange of device tensors")
            if dist.is_initialized() and (info["n_ranks"], info["rank"]) != (self.world, self.rank):
                raise ValueError(f"comm: the communicator is rank {info['rank']} of {info['n_ranks']}, "
                                 f"the group rank {self.rank} of {self.world}")
            self.world, self.rank = info["n_ranks"], info["rank"]
        self.collectives = self.world > 1 or (force_collectives and (dist.is_initialized() or comm is not None))
        # device tensors over RCCL: the side-stream pipeline described above
        self.on_side = self.collectives and torch.device(device).type == "cuda" and \
            (comm is not None or dist.get_backend(group) == "nccl")
        self.slabs = [torch.zeros((n_frames, slab_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.sparse = compact is not None
        if self.sparse and (place_tiles is None or not capacity):
            raise ValueError("the sparse exchange needs compact, place_tiles and the capacity in tiles")
        self.compact, self.place_tiles, self.entry_bytes = compact, place_tiles, entry_bytes
        if self.sparse:
            self.entries = [torch.empty((capacity, entry_bytes), dtype=torch.uint8, device=device) for _ in range(3)]
            self.count = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(3)]
            self.capacity = int(capacity)
            self.gathered = [None, None]
            if self.on_side:
                # High priority: HIP gives such a stream a hardware queue of its own class.  A stream of normal priority can be
                # handed the hardware queue the render stream uses (GPU_MAX_HW_QUEUES = 4, dealt round-robin) — then gather and
                # placement queue up BEHIND the renders instead of running beside them (seen in a kernel trace: both on queue 4,
                # strictly one after the other).  And a placement is short, memory-bound work that should not wait for a
                # persistent shading kernel to drain.
                # (``side_stream``: exchanges that share a communicator share the side stream too, so that the collectives reach
                # the communicator in ONE stream's order)
                self.side = side_stream if side_stream is not None else \
                    torch.cuda.Stream(device=device, priority=int(os.environ.get("RT_EXCHANGE_STREAM_PRIORITY", "-1")))
                self.counts_all = [torch.zeros(self.world, dtype=torch.int32, device=device) for _ in range(3)]
                self.counts_host = [torch.zeros(self.world, dtype=torch.int32).pin_memory() for _ in range(3)]
                self.reusable = [None, None, None]   # event on the side stream: the gather that read entries[b] is done
        else:
            self.gathered = [torch.empty((self.world, n_frames, slab_bytes), dtype=torch.uint8, device=device)
                             for _ in range(2)] if (self.rank == dst and self.world > 1) else [None, None]
        self.pending = [None, None]      # dense: (work handle, step) of the batch in flight in each buffer
        self.stage = {}                  # sparse: step -> state of a batch between its compaction and its placement
        self.sent_bytes = 0              # bytes this rank handed to the gather so far (either path)

    def slab(self, step: int):
        if not self.sparse:
            self._finish(step & 1)       # the batch that used this buffer two steps ago must be gone
        return self.slabs[step & 1]      # (sparse: compact(step - 2) was queued before this render on the same stream)

    # ---- dense -------------------------------------------------------------------------------
    def submit(self, step: int):
        if self.sparse:
            return self._submit_sparse(step)
        b = step & 1
        if self.world == 1:
            self.place(self.slabs[b].unsqueeze(0), step)
            return
        self.sent_bytes += self.slabs[b].numel()
        if _staged(self.slabs[b], self.dist, self.group):
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

    def _finish(self, b: int):
        if self.pending[b] is None:
            return
        work, step = self.pending[b]
        self.pending[b] = None
        if work is not None:
            work.wait()                  # nccl: the current stream waits; gloo: the host does
        if self.rank == self.dst:
            self.place(self.gathered[b], step)

    # ---- sparse ------------------------------------------------------------------------------
    def _submit_sparse(self, step: int):
        import torch
        b3 = step % 3
        assert step - 3 not in self.stage, "a batch's entries are still in flight: drain() or submit in order"
        dist, world = self.dist, self.world
        if self.on_side and self.reusable[b3] is not None:
            torch.cuda.current_stream().wait_event(self.reusable[b3])   # the gather of batch step - 3 has read entries[b3]
            self.reusable[b3] = None
        if self.contributes:
            self.compact(self.slabs[step & 1], self.entries[b3], self.count[b3])
        else:
            self.count[b3].zero_()               # (a rank that rendered nothing — a root that only places — joins with no entries)
        st = dict(b3=b3, counts=None, work=None, nmax=None, gathered=None, ready=None)
        if not self.collectives:
            st["counts"] = self.count[b3]
        elif self.on_side:
            st["counts"] = self.counts_all[b3]
            if self.comm is not None:
                packed = torch.cuda.Event()
                packed.record()                          # (render stream: this batch's compaction is queued)
            else:
                work = dist.all_gather_into_tensor(st["counts"], self.count[b3], group=self.group, async_op=True)
            with torch.cuda.stream(self.side):
                if self.comm is not None:
                    self.side.wait_event(packed)
                    self.comm.comm_all_gather_counts(self.count[b3], st["counts"])   # ncclAllGather on the side stream
                else:
                    work.wait()                          # the SIDE stream waits for the collective; the render stream goes on
                self.counts_host[b3].copy_(st["counts"], non_blocking=True)
                st["ready"] = torch.cuda.Event()
                st["ready"].record()
        elif _staged(self.count[b3], dist, self.group):
            every = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(every, self.count[b3].cpu(), group=self.group)
            st["counts"] = torch.cat(every).to(self.count[b3].device)
        else:
            st["counts"] = torch.empty(world, dtype=torch.int32, device=self.count[b3].device)
            st["work"] = dist.all_gather_into_tensor(st["counts"], self.count[b3], group=self.group, async_op=True) \
                if dist.get_backend(self.group) != "gloo" else \
                dist.all_gather(list(st["counts"].split(1)), self.count[b3], group=self.group, async_op=True)
        self.stage[step] = st
        self._gather_stage(step - 1)
        self._place_stage(step - 2)

    def _gather_stage(self, step: int):
        """The counts of batch `step` are in: one gather of [max count, entry_bytes] per rank."""
        import torch
        st = self.stage.get(step)
        if st is None or st["nmax"] is not None:
            return
        if st["work"] is not None:
            st["work"].wait()
            st["work"] = None
        if st["ready"] is not None:
            st["ready"].synchronize()                 # the host waits for this batch's count exchange, for nothing queued since
            counts = [int(c) for c in self.counts_host[st["b3"]].tolist()]
        else:
            counts = [int(c) for c in st["counts"].tolist()]
        if max(counts) > self.capacity:
            raise RuntimeError(f"sparse exchange: {max(counts)} occupied tiles exceed the capacity of {self.capacity}")
        nmax = st["nmax"] = max(counts)
        self.sent_bytes += (counts[self.rank] if self.comm is not None else nmax) * self.entry_bytes + 4
        send = self.entries[st["b3"]][:max(nmax, 1)]
        dist, world = self.dist, self.world
        if not self.collectives:
            st["gathered"] = send.unsqueeze(0)
        elif self.on_side:
            with torch.cuda.stream(self.side):        # (RCCL's stream waits for the stream the call is made on)
                if self.rank == self.dst:
                    st["gathered"] = torch.empty((world, max(nmax, 1), self.entry_bytes), dtype=torch.uint8, device=send.device)
                if self.comm is not None:             # grouped ncclSend / ncclRecv inside the library, on the side stream
                    eb = self.entry_bytes
                    self.comm.comm_gather(send, counts[self.rank] * eb, st["gathered"], [c * eb for c in counts],
                                          max(nmax, 1) * eb, self.dst)
                else:
                    st["work"] = dist.gather(send, list(st["gathered"].unbind(0)) if self.rank == self.dst else None,
                                             dst=self.dst, group=self.group, async_op=True)
        elif _staged(send, dist, self.group):
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if self.rank == self.dst else None
            dist.gather(host, parts, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                st["gathered"] = torch.stack(parts).to(send.device)
        else:
            if self.rank == self.dst:
                st["gathered"] = torch.empty((world, max(nmax, 1), self.entry_bytes), dtype=torch.uint8, device=send.device)
            st["work"] = dist.gather(send, list(st["gathered"].unbind(0)) if self.rank == self.dst else None,
                                     dst=self.dst, group=self.group, async_op=True)

    def _place_stage(self, step: int):
        st = self.stage.get(step)
        if st is None:
            return
        self._gather_stage(step)
        if self.on_side:
            import torch
            behind = self.rank == self.dst and not self.place_beside
            if self.defer_place and self.rank == self.dst:
                with torch.cuda.stream(self.side):
                    if st["work"] is not None:
                        st["work"].wait()
                    counts = st["counts"].clone()     # (counts_all[b3] is written again three batches on)
                    done = torch.cuda.Event()
                    done.record()
                self.reusable[st["b3"]] = done        # entries[b3] is free once the gather has read it
                self.deferred.append((st["gathered"], counts, step, done))
                del self.stage[step]
                return
            with torch.cuda.stream(self.side):
                if st["work"] is not None:
                    st["work"].wait()
                if self.rank == self.dst and not behind:
                    self.place_tiles(st["gathered"], st["counts"], step)
                # entries[b3] (read by the gather) and counts_all[b3] (read by the placement) may be written again after this
                done = torch.cuda.Event()
                done.record()
            if behind:                                # on the render stream, once the gather has landed
                main = torch.cuda.current_stream()
                main.wait_event(done)
                st["gathered"].record_stream(main)    # (allocated on the side stream)
                self.place_tiles(st["gathered"], st["counts"], step)
                done = torch.cuda.Event()
                done.record()
            self.reusable[st["b3"]] = done
            del self.stage[step]
            return
        if st["work"] is not None:
            st["work"].wait()
        del self.stage[step]
        if self.rank == self.dst:
            if self.defer_place:
                self.deferred.append((st["gathered"], st["counts"].clone(), step, None))
            else:
                self.place_tiles(st["gathered"], st["counts"], step)

    def place_pending(self):
        """defer_place: place every batch whose gather has been issued, in order, on the CURRENT stream (which waits for each
        gather to land).  Returns how many."""
        if not self.deferred:
            return 0
        import torch
        n = len(self.deferred)
        for gathered, counts, step, done in self.deferred:
            if done is not None:
                main = torch.cuda.current_stream()
                main.wait_event(done)
                gathered.record_stream(main)          # (both were allocated on the side stream)
                counts.record_stream(main)
            self.place_tiles(gathered, counts, step)
        self.deferred = []
        return n

    def drain(self):
        if self.sparse:
            for step in sorted(self.stage):
                self._place_stage(step)
            self.place_pending()
            if self.on_side:
                import torch
                torch.cuda.current_stream().wait_stream(self.side)   # what follows on the render stream sees the frames
            return
        # in submission order
        order = sorted((p[1], b) for b, p in enumerate(self.pending) if p is not None)
        for _, b in order:
            self._finish(b)


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


# ---- frame-parallel animation batches (BASELINE config 5) ------------------------------------------------------
def frames_of_rank(n_frames: int, rank: int, world: int) -> np.ndarray:
    """Frame-parallel animation (config 5): frame f -> rank f mod G."""
    return np.arange(rank, n_frames, world)


def render_frames_parallel(renderer, cameras, seeds, width, height, samples=1, *, group=None, out=None, **kw):
    """Config 5: this rank renders frames rank, rank + G, ... of the animation WHOLE (``render`` called per frame,
    raytracer.py:310-325 / :334, batched into launches); no exchange until ``gather_frames``.  ``cameras`` may be
    the full list (indexed here) or a packed rt_camera array of this rank's frames.  Returns (uint8 tensor
    [n_local, H, W, 3] on this rank's GPU, the frame numbers it holds)."""
    import ctypes
    import torch.distributed as dist
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    seeds = list(seeds)
    mine = frames_of_rank(len(seeds), rank, world)
    cams = cameras if isinstance(cameras, ctypes.Array) else [cameras[int(f)] for f in mine]
    local = renderer.render_frames(cams, [seeds[int(f)] for f in mine], width, height, samples, out=out, **kw)
    return local, mine


def gather_frames(local, n_frames: int, group=None, dst: int = 0, out=None):
    """The one exchange of a frame-parallel batch: every rank's frames to ``dst`` -> [n_frames, H, W, 3] in frame
    order (None elsewhere).  One fixed-size gather of ceil(n_frames / G) frames per rank (ranks with one frame
    fewer pad)."""
    import torch
    import torch.distributed as dist
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    shape = tuple(local.shape[1:])
    if out is None and rank == dst:
        out = torch.empty((n_frames,) + shape, dtype=local.dtype, device=local.device)
    if world == 1:
        out.copy_(local)
        return out
    n_max = -(-n_frames // world)
    send = local
    if local.shape[0] < n_max:
        send = torch.zeros((n_max,) + shape, dtype=local.dtype, device=local.device)
        send[:local.shape[0]].copy_(local)
    if _staged(send, dist, group):
        host = send.cpu()
        parts = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
        dist.gather(host, parts, dst=dst, group=group)
        gathered = torch.stack(parts).to(local.device) if rank == dst else None
    else:
        gathered = torch.empty((world, n_max) + shape, dtype=local.dtype, device=local.device) if rank == dst else None
        dist.gather(send.contiguous(), list(gathered.unbind(0)) if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None
    for r in range(world):
        k = len(frames_of_rank(n_frames, r, world))
        if k:
            out[r::world].copy_(gathered[r, :k])
    return out
