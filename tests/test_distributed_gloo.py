"""The N>1 exchange step on CPU: world_size-2 (and 3) gloo runs of gather_row_tiles.  Each rank's
rows come from the oracle here (no GPU in this container); on the GPU box the same function moves
rt_render_rows output over RCCL (tests/test_gpu_properties.py covers tile == full-frame)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, w, h, tile_rows, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cpu_oracle
        from raytracer_amd import tiling
        from raytracer_amd.distributed import frames_of_rank, gather_row_tiles
        from raytracer_amd.scenes import default_scene
        rows = tiling.local_rows(h, tile_rows, rank, world)
        px = np.stack([np.tile(np.arange(w), len(rows)), np.repeat(rows, w)], 1)
        o = cpu_oracle.render_scene(default_scene(), w, h, pixels=px, max_depth=3, mode="K", seed=4, threads=2)
        local = torch.from_numpy(o["rgb8"].reshape(len(rows), w, 3))
        img = gather_row_tiles(local, w, h, tile_rows)
        if rank == 0:
            full = cpu_oracle.render_scene(default_scene(), w, h, max_depth=3, mode="K", seed=4, threads=2)
            ok = np.array_equal(img.numpy(), full["rgb8"].reshape(h, w, 3))
            fr = [frames_of_rank(10, r, world).tolist() for r in range(world)]
            q.put((ok, bool(img.numpy().any()), sorted(sum(fr, [])) == list(range(10))))
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile_rows", [(2, 64, 48, 8), (3, 40, 37, 8), (2, 16, 5, 8)])
def test_gather_row_tiles_gloo(world, w, h, tile_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, tile_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, nonblack, frames_ok = q.get(timeout=5)
    assert ok and frames_ok and (nonblack or h < 8)


def _sparse_exchange_worker(rank, world, port, q):
    """Mostly black slabs with a different number of lit segments on every rank and step (also none at all)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd.distributed import BatchExchange
        F, nbytes, steps = 4, 96 * 5, 6

        def slab_of(step, r):
            g = torch.Generator().manual_seed(1000 * step + r)
            t = torch.zeros(F * nbytes, dtype=torch.uint8)
            k = 0 if (step == 2 or (step == 4 and r == 0)) else int(torch.randint(1, 40, (1,), generator=g))
            pos = torch.randint(0, F * nbytes, (k,), generator=g)
            t[pos] = torch.randint(1, 256, (k,), generator=g).to(torch.uint8)
            if step == 5:
                t[-1] = 7; t[0] = 9          # first and last segment
            return t.view(F, nbytes)

        placed = []

        def place(gathered, step):
            exp = torch.stack([slab_of(step, r) for r in range(world)])
            placed.append((step, bool(torch.equal(gathered, exp))))

        ex = BatchExchange(F, nbytes, "cpu", place, sparse=True)
        assert ex.sparse
        for i in range(steps):
            ex.slab(i).copy_(slab_of(i, rank))
            ex.submit(i)
        ex.drain()
        assert ex.sent_bytes < steps * F * nbytes / 2          # (dense: steps * F * nbytes)
        assert not BatchExchange(3, 40, "cpu", place, sparse=True).sparse   # 120 bytes: not a multiple of 32 -> dense
        if rank == 0:
            q.put(placed)
    finally:
        dist.destroy_process_group()


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd.distributed import BatchExchange
        F, nbytes, steps = 3, 40, 7
        placed = []

        def place(gathered, step):       # root only: every rank's slab of that step, in submission order
            exp = torch.stack([torch.full((F, nbytes), (step * 16 + r) % 251, dtype=torch.uint8) for r in range(world)])
            placed.append((step, bool(torch.equal(gathered, exp))))

        ex = BatchExchange(F, nbytes, "cpu", place)
        for i in range(steps):
            ex.slab(i).fill_((i * 16 + rank) % 251)       # "render" this rank's tiles of batch i
            ex.submit(i)
        ex.drain()
        if rank == 0:
            q.put(placed)
        else:
            assert placed == []
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batch_exchange_pipeline_gloo(world):
    """bench.py's N > 1 loop: double-buffered gather of frame batches overlapped with the next render."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    placed = q.get(timeout=5)
    assert [s for s, _ in placed] == list(range(7)) and all(ok for _, ok in placed)


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_batch_exchange_gloo(world):
    """BatchExchange(sparse=True): only non-black 32-byte segments travel; the root sees the same bytes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    placed = q.get(timeout=5)
    assert [s for s, _ in placed] == list(range(6)) and all(ok for _, ok in placed)


def test_segments_to_frames_matches_row_placement():
    """segments_to_frames puts the 32-byte segments of every rank's slab where the dense path's row placement
    puts the rows (CPU tensors; ragged last tile, ranks with fewer rows than rows_max)."""
    from raytracer_amd import tiling
    from raytracer_amd.distributed import segments_to_frames
    w, h, tr, F = 32, 45, 8, 3
    for world in (2, 3, 4):
        rows_max = tiling.max_local_rows(h, tr, world)
        g = torch.Generator().manual_seed(world)
        frames = torch.zeros((F, h, w, 3), dtype=torch.uint8)
        lit = torch.rand((F, h, w), generator=g) < 0.05
        frames[lit] = torch.randint(1, 256, (int(lit.sum()), 3), generator=g).to(torch.uint8)
        rows_all, counts = [], []
        for r in range(world):
            lr = torch.from_numpy(tiling.local_rows(h, tr, r, world).astype("int64"))
            slab = torch.zeros((F, rows_max, w * 3), dtype=torch.uint8)
            slab[:, :len(lr)] = frames[:, lr].reshape(F, len(lr), w * 3)
            seg = slab.view(-1).view(torch.int64).view(-1, 4)
            nz = (seg != 0).any(1).nonzero().squeeze(1)
            rows_all.append(torch.cat([nz[:, None], seg[nz]], 1)); counts.append(len(nz))
        nmax = max(counts)
        rows = torch.zeros((world, nmax, 5), dtype=torch.int64)
        for r in range(world):
            rows[r, :counts[r]] = rows_all[r]
        out = torch.full((F, h, w, 3), 77, dtype=torch.uint8)
        assert torch.equal(segments_to_frames(rows, counts, out, w, h, tr, rows_max), frames)


def test_batch_exchange_single_process():
    from raytracer_amd.distributed import BatchExchange
    seen = []
    ex = BatchExchange(2, 8, "cpu", lambda g, i: seen.append((i, g.shape, int(g[0, 0, 0]))))
    for i in range(3):
        ex.slab(i).fill_(i + 1)
        ex.submit(i)
    ex.drain()
    assert seen == [(0, (1, 2, 8), 1), (1, (1, 2, 8), 2), (2, (1, 2, 8), 3)]


def _gpu_exchange_worker(rank, world, port, q, sparse=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import tiling
        from raytracer_amd.distributed import BatchExchange
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera
        w, h, tr, F, steps = 320, 187, 8, (4 if sparse else 5), 3     # (sparse: batch bytes a multiple of 32)
        r = Renderer(0)
        r.upload_scene(generate_scene(8, 1, keep_default=True))
        rows_max = tiling.max_local_rows(h, tr, world)
        frames_out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
        ok = []

        def place(gathered, i):          # rank 0: assemble batch i and compare it with whole-frame renders
            r.scatter_rows_frames(gathered, world, w, h, tr, out=frames_out)
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            ok.append(bool(torch.equal(frames_out, full)))

        def place_segments(rows, counts, i):   # sparse, no dense intermediate: segments straight into image order
            from raytracer_amd.distributed import segments_to_frames
            segments_to_frames(rows, counts, frames_out, w, h, tr, rows_max)
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            ok.append(bool(torch.equal(frames_out, full)))

        ex = BatchExchange(F, rows_max * w * 3, "cuda:0", place, sparse=bool(sparse),
                           place_segments=place_segments if sparse == 2 else None)
        assert ex.sparse == bool(sparse)
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=ex.slab(i).view(F, rows_max, w, 3),
                            tile_rows=tr, tile_first=rank, tile_stride=world)
            ex.submit(i)
        ex.drain()
        if rank == 0:
            q.put(ok)
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [0, 1, 2])     # dense / sparse via dense buffer / sparse straight into the frames
@pytest.mark.parametrize("world", [2, 3])
def test_batch_exchange_of_gpu_row_tiles(world, sparse):
    """bench.py's N > 1 step on the GPU: `world` processes (all on cuda:0 here) each render their row tiles
    of a batch with the HIP library, the slabs are gathered (gloo, staged through host memory — RCCL
    needs one GPU per rank) and rank 0 places them with rt_scatter_rows_frames: the assembled frames
    are byte-identical to whole-frame renders."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_exchange_worker, args=(r, world, port, q, sparse)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    ok = q.get(timeout=5)
    assert ok == [True, True, True]
