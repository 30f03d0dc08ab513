"""First contact with RCCL: torch.distributed backend "nccl" (= RCCL on ROCm) as a group of ONE rank on the GPU box.
One rank needs no second GPU, but every call goes through ProcessGroupNCCL and librccl with device tensors: the exact
collectives BatchExchange makes (all_gather_into_tensor of the counts, asynchronous gather into unbind(0) views, barrier),
and the whole side-stream pipeline of the sparse exchange (force_collectives) around real renders — compaction, count
exchange, pinned read-back, gather, placement on the side stream — assembled frames byte-identical to whole-frame
renders.  Several ranks meet gloo only (tests/test_distributed_gloo.py); a multi-GPU node is the driver's to run.

The worker runs in a process of its own: a process group must not outlive the test inside pytest's process."""
import multiprocessing as mp
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    res = {}
    try:
        dev = "cuda:0"
        # ---- the calls themselves
        mine = torch.tensor([41], dtype=torch.int32, device=dev)
        every = torch.zeros(1, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(every, mine, async_op=True).wait()
        res["all_gather_into_tensor"] = int(every.item()) == 41
        send = torch.arange(5 * 196, dtype=torch.int32, device=dev).to(torch.uint8).view(5, 196)
        gathered = torch.zeros((1, 5, 196), dtype=torch.uint8, device=dev)
        dist.gather(send, list(gathered.unbind(0)), dst=0, async_op=True).wait()
        res["gather_into_views"] = bool(torch.equal(gathered[0], send))
        t = torch.ones(4, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        res["all_reduce_barrier"] = bool((t == 1).all())
        res["backend"] = dist.get_backend()

        # ---- the sparse exchange as bench.py drives it at N > 1, as a group of one
        from raytracer_amd import tiling
        from raytracer_amd.distributed import BatchExchange, gather_frames
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera
        w, h, tr, F, steps = 320, 187, 8, 6, 7
        r = Renderer(0)
        r.upload_scene(generate_scene(8, 1, keep_default=True))
        rows_max = tiling.max_local_rows(h, tr, 1)
        frames_out = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
        kept = {}

        def place_tiles(entries, counts, i):
            r.place_tiles(entries, counts, w, h, tr, frames_out, on_current_stream=True)
            kept[i] = frames_out.clone()         # (on the side stream, after the placement)

        ex = BatchExchange(F, rows_max * w * 3, dev, None, compact=r.compact_tiles, place_tiles=place_tiles,
                           capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), force_collectives=True)
        res["on_side"] = bool(ex.on_side)
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=ex.slab(i).view(F, rows_max, w, 3),
                            tile_rows=tr, tile_first=0, tile_stride=1)
            ex.submit(i)
        ex.drain()
        torch.cuda.synchronize()
        ok = []
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            ok.append(bool(torch.equal(kept[i], full)) and bool(full.any()))
        res["exchange"] = ok
        res["sent_bytes"] = int(ex.sent_bytes)
        # ---- the same with deferred placement: the pipeline gathers, the caller places (every third batch here, and at drain)
        kept.clear()
        exd = BatchExchange(F, rows_max * w * 3, dev, None, compact=r.compact_tiles, place_tiles=place_tiles,
                            capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), force_collectives=True, defer_place=True)
        placed_at = []
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=exd.slab(i).view(F, rows_max, w, 3),
                            tile_rows=tr, tile_first=0, tile_stride=1)
            exd.submit(i)
            if i % 3 == 2:
                placed_at.append(exd.place_pending())
        res["deferred_nothing_early"] = len(kept) == sum(placed_at)
        exd.drain()
        torch.cuda.synchronize()
        okd = []
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            okd.append(i in kept and bool(torch.equal(kept[i], full)))
        res["deferred_exchange"] = okd
        res["deferred_batches"] = placed_at
        # ---- config 5's one gather at the end (world 1: a copy), and the dense exchange's gather
        res["gather_frames"] = bool(torch.equal(gather_frames(full, F), full))
        seen = []
        exd = BatchExchange(F, rows_max * w * 3, dev, lambda g, i: seen.append((i, tuple(g.shape))))
        exd.slab(0); exd.submit(0); exd.drain()
        res["dense_world1"] = seen == [(0, (1, F, rows_max * w * 3))]
        r.close()
    except Exception as e:      # noqa: BLE001  (reported to the parent)
        res["error"] = repr(e)
    finally:
        dist.destroy_process_group()
        q.put(res)


def test_rccl_group_of_one_runs_the_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert "error" not in res, res
    assert res["backend"] == "nccl" and res["on_side"]
    assert res["all_gather_into_tensor"] and res["gather_into_views"] and res["all_reduce_barrier"]
    assert res["exchange"] == [True] * 7 and res["sent_bytes"] > 0
    assert res["deferred_nothing_early"] and res["deferred_exchange"] == [True] * 7 and sum(res["deferred_batches"]) >= 3, res
    assert res["gather_frames"] and res["dense_world1"]
    assert p.exitcode == 0


def _worker_abi(q):
    """The library's own communicator (rt_comm_*, include/rt355.h) as a group of ONE rank: no torch.distributed at all."""
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    res = {}
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # (a hang ends here, with the stack on stderr, not at the harness's limit)
    try:
        from raytracer_amd import _lib, tiling
        from raytracer_amd.distributed import BatchExchange
        from raytracer_amd.render import Renderer
        from raytracer_amd.scenes import generate_scene, orbit_camera
        dev = "cuda:0"
        L = _lib.load()
        r = Renderer(0)
        # ---- refusals before a communicator exists, and bad arguments
        one = torch.tensor([41], dtype=torch.int32, device=dev)
        every = torch.zeros(1, dtype=torch.int32, device=dev)
        res["no_comm"] = (L.rt_comm_all_gather_counts(r._h, one.data_ptr(), every.data_ptr(), None) == -3 and
                          L.rt_comm_gather(r._h, one.data_ptr(), 4, every.data_ptr(), (ctypes.c_int64 * 1)(4), 4, 0, None) == -3)
        res["info_before"] = r.comm_info()
        ident = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
        res["bad_init"] = (L.rt_comm_init(r._h, ident, 0, 0) == -1 and L.rt_comm_init(r._h, ident, 2, 2) == -1 and
                           L.rt_comm_init(r._h, None, 1, 0) == -1 and L.rt_comm_unique_id(None) == -1)
        # ---- the calls themselves
        r.comm_init_single()
        res["info"] = r.comm_info()
        res["double_init"] = L.rt_comm_init(r._h, ident, 1, 0) == -3
        r.comm_all_gather_counts(one, every)
        send = torch.arange(7 * 196, dtype=torch.int32, device=dev).to(torch.uint8).view(7, 196)
        gathered = torch.zeros((1, 9, 196), dtype=torch.uint8, device=dev)
        r.comm_gather(send, 5 * 196, gathered, [5 * 196], 9 * 196, 0)          # 5 of the 7 entries travel
        torch.cuda.synchronize()
        res["all_gather_counts"] = int(every.item()) == 41
        res["gather"] = bool(torch.equal(gathered[0, :5], send[:5])) and not bool(gathered[0, 5:].any())
        res["bad_gather"] = (L.rt_comm_gather(r._h, send.data_ptr(), 196, gathered.data_ptr(), (ctypes.c_int64 * 1)(392), 9 * 196, 0, None) == -1 and
                             L.rt_comm_gather(r._h, send.data_ptr(), 196, gathered.data_ptr(), (ctypes.c_int64 * 1)(196), 100, 0, None) == -1 and
                             L.rt_comm_gather(r._h, send.data_ptr(), 196, gathered.data_ptr(), None, 196, 0, None) == -1 and
                             L.rt_comm_gather(r._h, send.data_ptr(), 196, gathered.data_ptr(), (ctypes.c_int64 * 1)(196), 196, 1, None) == -1)
        r.comm_gather(send, 0, gathered, [0], 9 * 196, 0)                        # nothing to send: nothing is posted
        torch.cuda.synchronize()
        # ---- the sparse exchange on the library's collectives (side-stream pipeline around real renders)
        w, h, tr, F, steps = 320, 187, 8, 6, 7
        r.upload_scene(generate_scene(8, 1, keep_default=True))
        rows_max = tiling.max_local_rows(h, tr, 1)
        frames_out = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
        kept = {}

        def place_tiles(entries, counts, i):
            r.place_tiles(entries, counts, w, h, tr, frames_out, on_current_stream=True)
            kept[i] = frames_out.clone()

        ex = BatchExchange(F, rows_max * w * 3, dev, None, compact=r.compact_tiles, place_tiles=place_tiles,
                           capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), force_collectives=True, comm=r)
        res["on_side"] = bool(ex.on_side) and ex.collectives
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=ex.slab(i).view(F, rows_max, w, 3),
                            tile_rows=tr, tile_first=0, tile_stride=1)
            ex.submit(i)
        ex.drain()
        torch.cuda.synchronize()
        ok = []
        for i in range(steps):
            cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
            full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
            ok.append(bool(torch.equal(kept[i], full)) and bool(full.any()))
        res["exchange"] = ok
        res["sent_bytes"] = int(ex.sent_bytes)
        # ---- deferred placement on the library's collectives: the pipeline gathers, the caller places (every third batch and
        # at drain) on ITS stream — a stream of its own here: on HIP's legacy default stream this very sequence deadlocks (the
        # default stream waits for an event behind ncclSend / ncclRecv), which is why bench.py renders on a stream of its own
        # whenever it exchanges
        kept.clear()
        with torch.cuda.stream(torch.cuda.Stream()):
            r.use_torch_stream()
            # (TWO exchanges on the one communicator, taking batches in turn and sharing the side stream: bench.py's layouts)
            exds = []
            for e in range(2):
                def place_e(entries, counts, k, e=e):
                    place_tiles(entries, counts, 2 * k + e)
                exds.append(BatchExchange(F, rows_max * w * 3, dev, None, compact=r.compact_tiles, place_tiles=place_e,
                                          capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8), force_collectives=True, comm=r,
                                          defer_place=True, side_stream=exds[0].side if exds else None))
            placed_at = []
            for i in range(steps):
                exd, k = exds[i & 1], i >> 1
                cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
                r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1, out=exd.slab(k).view(F, rows_max, w, 3),
                                tile_rows=tr, tile_first=0, tile_stride=1)
                exd.submit(k)
                if i % 3 == 2:
                    placed_at.append(sum(x.place_pending() for x in exds))
            res["deferred_nothing_early"] = len(kept) == sum(placed_at)
            for x in exds:
                x.drain()
            torch.cuda.synchronize()
            okd = []
            for i in range(steps):
                cams = [orbit_camera(10 * i + f, 64) for f in range(F)]
                full = r.render_frames(cams, range(i * F, (i + 1) * F), w, h, 1)
                okd.append(i in kept and bool(torch.equal(kept[i], full)))
        r.use_torch_stream()
        res["deferred_exchange"] = okd
        res["deferred_batches"] = placed_at
        r.comm_destroy()
        res["info_after"] = r.comm_info()
        r.comm_init_single()          # a context may join again; rt_destroy releases what it still holds
        r.close()
    except Exception as e:      # noqa: BLE001  (reported to the parent)
        res["error"] = repr(e)
    finally:
        q.put(res)


def test_library_communicator_group_of_one():
    """rt_comm_* through the C ABI on the GPU box: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / grouped
    ncclSend + ncclRecv (to itself) / ncclCommDestroy inside librt355, bound to librccl at run time."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_abi, args=(q,))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert "error" not in res, res
    assert res["no_comm"] and res["bad_init"] and res["double_init"] and res["bad_gather"], res
    assert res["info_before"]["n_ranks"] == 0 and res["info_before"]["rank"] == -1
    assert res["info"]["n_ranks"] == 1 and res["info"]["rank"] == 0 and res["info"]["rccl_version"] >= 20700, res["info"]
    assert res["all_gather_counts"] and res["gather"], res
    assert res["on_side"] and res["exchange"] == [True] * 7 and res["sent_bytes"] > 0, res
    assert res["deferred_nothing_early"] and res["deferred_exchange"] == [True] * 7 and sum(res["deferred_batches"]) >= 1, res
    assert res["info_after"]["n_ranks"] == 0
    assert p.exitcode == 0
