"""GPU box, ONE GPU: the ROOT's side of bench.py's layouts taking turns (RT_BENCH_LAYOUTS), as a pipeline — in A batches its
render share (tile_stride = N) and compaction on the render stream, in B batches nothing there for as long as a renderer's
B batch takes (a spin kernel stands in for the wait), and the placement of EVERY batch on a high-priority stream of its own.
Prints the time per pattern against what the renderers need for it: if the two agree, the root's placements do wait for its idle
batches.  (No RCCL: the gather's kernels need another GPU.)      python tools/pattern_emulation.py [N pattern] ..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from raytracer_amd import tiling
from raytracer_amd._lib import TILE_ENTRY_BYTES
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera

r = Renderer(0)
w, h, n, _, _, D, desc = bench.WORKLOADS["c2"]
F, tr = 256, 8
r.upload_scene(bench.build_scene("c2"))
cams = r.pack_cameras([Camera()] * F)
# RT_PLACE_PRIORITY: -1 = high (a hardware queue of its own class), 0 = normal, least = whatever Stream.priority_range() gives
_prio = os.environ.get("RT_PLACE_PRIORITY", "-1")
_range = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
place_stream = torch.cuda.Stream(priority=max(_range) if _prio == "least" else int(_prio))
print(f"priority range {_range}, placements at {place_stream.priority}", flush=True)
cases = [(int(a), b) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(2, "AAAAAB"), (3, "AAB"), (4, "AB")]
clock_hz = torch.cuda.get_device_properties(0).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


print("| N | pattern | renderer: A batch ms | B batch ms | per pattern ms | root alone: renders + compactions | placements | root measured per pattern ms | "
      "its render stream alone per pattern ms |")
print("|---|---|---|---|---|---|---|---|---|")
for N, pattern in cases:
    def setup(G):
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
        def render():
            r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=0, tile_stride=G)
            r.compact_tiles(slab, own, own_count)
        return dict(G=G, render=render, gathered=gathered, counts=counts, table={})
    lay = {"A": setup(N), "B": setup(N - 1)}
    out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
    tA, tB = timed(lay["A"]["render"]), timed(lay["B"]["render"])
    def place(L, on_stream):
        L["table"]["t"] = r.place_tiles(L["gathered"], L["counts"], w, h, tr, out, L["table"].get("t"), on_current_stream=on_stream)
    tP = {c: timed(lambda c=c: place(lay[c], False)) for c in "AB"}
    spin = int(tB * 1e-3 * clock_hz)

    def run(patterns, with_place):
        done = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(patterns):
            for c in pattern:
                if c == "A":
                    lay["A"]["render"]()
                else:
                    torch.cuda._sleep(spin)             # the renderers' B batch: nothing for the root's render stream to do
                if with_place:
                    ev = torch.cuda.Event(); ev.record()
                    with torch.cuda.stream(place_stream):
                        place_stream.wait_event(ev)
                        place(lay[c], True)
                        d = torch.cuda.Event(); d.record()
                    done.append(d)
                    if len(done) > 8:
                        torch.cuda.current_stream().wait_event(done.pop(0))     # (at most eight placements behind)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / patterns
    run(2, True)
    nA, nB = pattern.count("A"), pattern.count("B")
    print(f"| {N} | {pattern} | {tA:.3f} | {tB:.3f} | {nA * tA + nB * tB:.3f} | {nA * tA:.3f} | {sum(tP[c] for c in pattern):.3f} | "
          f"{run(10, True):.3f} | {run(10, False):.3f} |", flush=True)
