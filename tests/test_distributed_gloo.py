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


def _sparse_exchange_worker(rank, world, port, q, root_places_only=False, defer=False):
    """BatchExchange's sparse pipeline on CPU tensors: mostly black slabs with a different number of lit tiles on
    every rank and step (also none at all); compact / place_tiles are the numpy stand-ins of tests/tile_exchange_ref
    (the HIP entry points need a GPU: tests below)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tile_exchange_ref as ref
        from raytracer_amd import tiling
        from raytracer_amd.distributed import BatchExchange
        F, w, h, tr, steps = 3, 40, 37, 8, 7
        # (root_places_only: bench.py's layout from three GPUs on — ranks 1 .. world - 1 render the row tiles of a group of
        # world - 1, rank 0 joins every collective with no entries and places the others')
        G, ridx = (world - 1, rank - 1) if root_places_only else (world, rank)
        rows_max = tiling.max_local_rows(h, tr, G)
        n_rows = tiling.local_row_count(h, tr, ridx, G) if ridx >= 0 else 0

        def frames_of(step):
            g = np.random.default_rng(100 + step)
            fr = np.zeros((F, h, w, 3), np.uint8)
            k = 0 if step == 2 else int(g.integers(1, 60))
            ys, xs, fs = g.integers(0, h, k), g.integers(0, w, k), g.integers(0, F, k)
            fr[fs, ys, xs] = g.integers(1, 256, (k, 3))
            if step == 5:
                fr[0, 0, 0] = 9; fr[-1, -1, -1] = 7          # first and last pixel
            return fr

        def slab_of(step, r):
            rows = tiling.local_rows(h, tr, r, G)
            sl = np.zeros((F, rows_max, w, 3), np.uint8)
            sl[:, :len(rows)] = frames_of(step)[:, rows]
            if step == 4 and r == 0:
                sl[:] = 0                                      # a rank with nothing to send
            return sl

        def compact(slab, entries, count):
            e, n = ref.compact(slab.view(F, rows_max, w, 3).numpy(), n_rows, w)
            entries[:n] = torch.from_numpy(e)
            count[0] = n

        placed = []

        def place_tiles(entries, counts, step):
            if root_places_only:
                assert int(counts[0]) == 0
            got = ref.place(entries[world - G:].numpy(), counts[world - G:].numpy(), F, w, h, tr)
            exp = frames_of(step)
            if step == 4:
                exp[:, tiling.local_rows(h, tr, 0, G)] = 0
            placed.append((step, bool(np.array_equal(got, exp))))

        ex = BatchExchange(F, rows_max * w * 3, "cpu", None, compact=compact, place_tiles=place_tiles,
                           capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), contributes=ridx >= 0,
                           defer_place=defer)
        assert ex.sparse
        for i in range(steps):
            if ridx >= 0:
                ex.slab(i).copy_(torch.from_numpy(slab_of(i, ridx)).view(F, -1))
            ex.submit(i)
            if defer:                    # (the pipeline gathers; nothing is placed until the caller asks)
                assert len(placed) == (i // 3) * 3 - 2 * (i >= 3) or rank != 0 or i < 3
                if i % 3 == 2:
                    ex.place_pending()
        ex.drain()
        assert ex.sent_bytes < steps * F * rows_max * w * 3 / 2      # (dense: the whole slabs)
        if rank == 0:
            q.put(placed)
        else:
            assert placed == []
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


@pytest.mark.parametrize("world,root_places_only,defer", [(2, False, False), (3, False, False), (3, True, False), (2, False, True),
                                                          (3, True, True)])
def test_sparse_batch_exchange_gloo(world, root_places_only, defer):
    """BatchExchange's sparse mode: only occupied tiles travel, three pipelined stages; the root assembles the
    same frames — also where it renders nothing itself (bench.py's layout B: the other ranks share the row tiles, rank 0
    joins with no entries), and with deferred placement (the caller says when: place_pending())."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_exchange_worker, args=(r, world, port, q, root_places_only, defer)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    placed = q.get(timeout=5)
    assert [s for s, _ in placed] == list(range(7)) and all(ok for _, ok in placed)


def test_batch_exchange_single_process():
    from raytracer_amd.distributed import BatchExchange
    seen = []
    ex = BatchExchange(2, 8, "cpu", lambda g, i: seen.append((i, g.shape, int(g[0, 0, 0]))))
    for i in range(3):
        ex.slab(i).fill_(i + 1)
        ex.submit(i)
    ex.drain()
    assert seen == [(0, (1, 2, 8), 1), (1, (1, 2, 8), 2), (2, (1, 2, 8), 3)]


def _gpu_exchange_worker(rank, world, port, q, sparse=False, glass=False, root_places_only=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import tiling
        from raytracer_amd.distributed import BatchExchange
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera, refractive_scene
        w, h, tr, F, steps = 320, 187, 8, 5, 4
        r = Renderer(0)
        r.upload_scene(refractive_scene() if glass else generate_scene(8, 1, keep_default=True))
        # (root_places_only: bench.py's layout from three GPUs on — ranks 1 .. world - 1 share the row tiles, rank 0 places)
        G, ridx = (world - 1, rank - 1) if root_places_only else (world, rank)
        rows_max = tiling.max_local_rows(h, tr, G)
        frames_out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
        ok = []

        def check(i):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            if glass:    # pixel sums are float atomics: equal up to summation order (<= 1/255 on a few pixels)
                d = (frames_out.to(torch.int16) - full.to(torch.int16)).abs()
                ok.append(int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01 and bool(full.any()))
            else:
                ok.append(bool(torch.equal(frames_out, full)))

        def place(gathered, i):          # rank 0: assemble batch i and compare it with whole-frame renders
            r.scatter_rows_frames(gathered[world - G:], G, w, h, tr, out=frames_out)
            check(i)

        def place_tiles(entries, counts, i):   # sparse: the HIP placement kernel (black + tiles in one pass)
            frames_out.fill_(77)
            r.place_tiles(entries[world - G:], counts[world - G:], w, h, tr, frames_out)
            check(i)

        ex = BatchExchange(F, rows_max * w * 3, "cuda:0", place, compact=r.compact_tiles, place_tiles=place_tiles,
                           capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), contributes=ridx >= 0) if sparse else \
            BatchExchange(F, rows_max * w * 3, "cuda:0", place)
        assert ex.sparse == bool(sparse)
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            if ridx >= 0:
                r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=ex.slab(i).view(F, rows_max, w, 3),
                                tile_rows=tr, tile_first=ridx, tile_stride=G)
            ex.submit(i)
        ex.drain()
        if rank == 0:
            q.put(ok)
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [0, 1])     # dense slabs / occupied tiles only (rt_compact_tiles + rt_place_tiles)
@pytest.mark.parametrize("world", [2, 3])
def test_batch_exchange_of_gpu_row_tiles(world, sparse):
    """bench.py's N > 1 step on the GPU: `world` processes (all on cuda:0 here) each render their row tiles
    of a batch with the HIP library, the slabs — or only their occupied tiles — are gathered (gloo, staged
    through host memory — RCCL needs one GPU per rank) and rank 0 places them with the HIP kernels: the
    assembled frames are byte-identical to whole-frame renders."""
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
    assert ok == [True] * 4


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [0, 1])
def test_batch_exchange_with_a_root_that_only_places(sparse):
    """bench.py's layout from three GPUs on: ranks 1 and 2 render the row tiles of a group of two, rank 0 renders nothing,
    joins the collectives (with no entries in the sparse exchange) and places the others' tiles: byte-identical frames."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_exchange_worker, args=(r, 3, port, q, sparse, False, True)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    ok = q.get(timeout=5)
    assert ok == [True] * 4


def _gpu_mixed_layouts_worker(rank, world, port, q, pattern):
    """bench.py's N > 1 loop with two layouts taking turns (A: all ranks render; B: the root only places), each with an
    exchange of its own: every batch must come out as the whole-frame render of its frames."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import tiling
        from raytracer_amd.distributed import BatchExchange
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera
        w, h, tr, F, steps = 320, 187, 8, 4, 7
        r = Renderer(0)
        r.upload_scene(generate_scene(8, 1, keep_default=True))
        ok = []
        lay = {}
        for letter in sorted(set(pattern)):
            G, ridx = (world, rank) if letter == "A" else (world - 1, rank - 1)
            L = dict(G=G, ridx=ridx, rows_max=tiling.max_local_rows(h, tr, G), count=0, batch_of={},
                     fo=torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0"))

            def place_tiles(entries, counts, k, L=L):
                L["fo"].fill_(77)
                r.place_tiles(entries[world - L["G"]:], counts[world - L["G"]:], w, h, tr, L["fo"])
                i = L["batch_of"][k]
                cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
                ok.append((i, bool(torch.equal(L["fo"], r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)))))

            L["ex"] = BatchExchange(F, L["rows_max"] * w * 3, "cuda:0", None, compact=r.compact_tiles, place_tiles=place_tiles,
                                    capacity=F * (((w + 7) // 8) | 1) * ((L["rows_max"] + 7) // 8), contributes=ridx >= 0,
                                    defer_place=True)
            lay[letter] = L
        for i in range(steps):
            L = lay[pattern[i % len(pattern)]]
            k = L["count"]; L["count"] += 1
            L["batch_of"][k] = i
            if L["ridx"] < 0:                    # (the root's idle batch: it places what has been gathered so far)
                before = len(ok)
                for M in lay.values():
                    M["ex"].place_pending()
                assert i < 5 or len(ok) > before      # (an exchange places two of its own submits behind)
            if L["ridx"] >= 0:
                cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
                r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=L["ex"].slab(k).view(F, L["rows_max"], w, 3),
                                tile_rows=tr, tile_first=L["ridx"], tile_stride=L["G"])
            L["ex"].submit(k)
        for L in lay.values():
            L["ex"].drain()
        if rank == 0:
            q.put(sorted(ok))
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,pattern", [(2, "AAB"), (3, "AB")])
def test_batch_exchange_with_layouts_taking_turns(world, pattern):
    """bench.py gives the root a part share of the rendering by letting two layouts take turns, batch by batch, each with
    its own exchange: all seven batches arrive byte-identical to whole-frame renders."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_mixed_layouts_worker, args=(r, world, port, q, pattern)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    ok = q.get(timeout=5)
    assert ok == [(i, True) for i in range(7)]


@pytest.mark.gpu
def test_batch_exchange_of_a_refractive_scene():
    """The same exchange on a scene with glass: a refractive frame is summed with float atomics, so the assembled
    batch equals a whole-frame render up to summation order — <= 1/255 on fewer than 1 % of the bytes — not byte for
    byte (two ranks, occupied tiles found by scanning: the visibility pass writes no pixels on this path)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_exchange_worker, args=(r, 2, port, q, 1, True)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    assert q.get(timeout=5) == [True] * 4


def _frame_parallel_worker(rank, world, port, q):
    """Config 5: frame f -> rank f mod G, whole frames, one gather at the end."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd.distributed import gather_frames, render_frames_parallel
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera
        w, h, F = 160, 90, 37                                # (37 frames: ranks hold different numbers of them)
        r = Renderer(0)
        r.upload_scene(generate_scene(8, 1, keep_default=True))
        cams = [orbit_camera(f, F) for f in range(F)]
        local, mine = render_frames_parallel(r, cams, range(F), w, h, 1, max_depth=5)
        assert list(mine) == list(range(rank, F, world)) and local.shape[0] == len(mine)
        frames = gather_frames(local, F)
        if rank == 0:
            full = r.render_frames(cams, range(F), w, h, 1, max_depth=5)
            q.put(bool(torch.equal(frames, full)) and bool(frames.any()))
        else:
            assert frames is None
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_frame_parallel_animation_gpu(world):
    """BASELINE config 5's mode end to end: every rank renders its frames of the orbit WHOLE; the gathered
    batch equals a single-rank render of all frames byte for byte."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_parallel_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _gather_frames_cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd.distributed import frames_of_rank, gather_frames
        F = 11
        mine = frames_of_rank(F, rank, world)
        local = torch.stack([torch.full((4, 6, 3), int(f) + 1, dtype=torch.uint8) for f in mine]) if len(mine) else \
            torch.zeros((0, 4, 6, 3), dtype=torch.uint8)
        out = gather_frames(local, F)
        if rank == 0:
            q.put([int(out[f, 0, 0, 0]) for f in range(F)] == list(range(1, F + 1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_frames_gloo(world):
    """The final exchange of a frame-parallel batch on CPU tensors: frame order restored, uneven shares."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_frames_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
