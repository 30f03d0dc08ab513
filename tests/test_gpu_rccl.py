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
    assert res["gather_frames"] and res["dense_world1"]
    assert p.exitcode == 0
