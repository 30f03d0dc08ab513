"""GPU box, ONE GPU: the ROOT's side of an N-GPU row-tile step, as a pipeline — its render share (tile_stride = N) and the
compaction on the render stream, rt_place_tiles of all N ranks' entries on a high-priority side stream beside the
next batches.  Period per batch of 256 frames; "on the render stream" = the placement queued behind the render.  (The gather's RCCL kernels are not part
of it: they need another GPU.)      python tools/root_emulation.py [workload] > profiles/r3_root_emulation.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from raytracer_amd import tiling
from raytracer_amd._lib import TILE_ENTRY_BYTES
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
r = Renderer(0)
w, h, n, _, _, D, desc = bench.WORKLOADS[wl]
F, tr = (16 if wl == "c4" else 256), 8
r.upload_scene(bench.build_scene(wl))
cams = r.pack_cameras([Camera()] * F)
side = torch.cuda.Stream(priority=-1)
print(f"## {wl}: {desc}; batches of {F} frames")
print("| N | render + compact alone ms | place alone ms | period, placement on the render stream ms | period, placement beside the renders ms |")
print("|---|---|---|---|---|")
for G in (2, 4, 8):
    rows_max = tiling.max_local_rows(h, tr, G)
    cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
    slab = torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
    entries = torch.empty((G, cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
    counts = torch.zeros(G, dtype=torch.int32, device="cuda:0")
    for rank in range(G):
        r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=rank, tile_stride=G)
        r.compact_tiles(slab, entries[rank], counts[rank:rank + 1])
    torch.cuda.synchronize()
    gathered = entries[:, :max(int(counts.max().item()), 1)].contiguous()
    own = torch.empty((cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
    own_count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
    table = {}

    def render():
        r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=0, tile_stride=G)
        r.compact_tiles(slab, own, own_count)

    def place(on_side):
        table["t"] = r.place_tiles(gathered, counts, w, h, tr, out, table.get("t"), on_current_stream=on_side)

    def run(iters, beside):
        done = [None, None]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            render()
            if beside:
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    place(True)
                    d = torch.cuda.Event(); d.record()
                if done[i & 1] is not None:
                    torch.cuda.current_stream().wait_event(done[i & 1])   # at most two placements behind
                done[i & 1] = d
            else:
                place(False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / iters

    def alone(fn, iters=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / iters

    run(5, True)
    print(f"| {G} | {alone(render):.3f} | {alone(lambda: place(False)):.3f} | {run(30, False):.3f} | {run(30, True):.3f} |", flush=True)
