"""GPU box, ONE GPU: the pieces of an N-GPU row-tile step measured one by one, and the scaling curve they predict
(DESIGN.md 7).  For each workload and N in 1, 2, 4, 8: rank 0's render share of a batch (tile_first = 0, tile_stride = N),
rt_compact_tiles of its slab, the bytes a rank hands to the gather, and rt_place_tiles of all N ranks' entries on the root
(every rank's share is rendered here, one after another, to have real entries).  Model of the steady state of the
pipelined exchange (raytracer_amd/distributed.py): a batch leaves every
    period(N) = max( render + compact [a rank that renders every batch],  bytes per rank / link rate [each peer has its own xGMI
                     link into GPU 0],  root: its renders + compactions + the placement of EVERY batch ) averaged over bench.py's
                     pattern of layouts (A: all N ranks render; B: ranks 1 .. N - 1 render, the root gathers and places only)
    python tools/scaling_model.py [workloads...] [--frames F (per batch; default 256, config 4: F / 16)] > profiles/r3_scaling_model.txt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from raytracer_amd import tiling
from raytracer_amd._lib import TILE_ENTRY_BYTES
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera

LINK_GBS = 153.0          # one xGMI link, one direction (MI355X_MICROARCH.md)
tr = 8
r = Renderer(0)


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


FRAMES = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 256
# bench.py's patterns of layouts (the root renders in some batches only); --plain: one layout throughout (A at N = 2, B from 3 on)
PATTERNS = None if "--plain" in sys.argv else {1: "A", 2: "AAAAAB", 3: "AAB", 4: "AB"}
workloads = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--frames"]
for wl in (workloads or ["c2", "t64", "c4"]):
    w, h, n, _, _, D, desc = bench.WORKLOADS[wl]
    F = max(1, FRAMES // 16) if wl == "c4" else FRAMES
    r.upload_scene(bench.build_scene(wl))
    cams = r.pack_cameras([Camera()] * F)
    print(f"## {wl}: {desc}; batches of {F} frames, row tiles of {tr} rows")
    print("| N | layouts by batch | render share ms A / B | compact ms A / B | MB per rank A / B | link ms @153 GB/s per batch | "
          "place ms (root) | renderers' work per batch ms | root's work per batch ms | bound | frames/s | speed-up |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    base = None

    def pieces(G, with_place):
        """Layout with G ranks rendering: the first renderer's share and compaction, MB per rank, placement of all G ranks' entries."""
        rows_max = tiling.max_local_rows(h, tr, G)
        cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
        slab = torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
        entries = torch.empty((G, cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
        counts = torch.zeros(G, dtype=torch.int32, device="cuda:0")
        t_render = t_compact = None
        for rank in range(G):
            def render():
                r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=rank, tile_stride=G)
            def compact():
                r.compact_tiles(slab, entries[rank], counts[rank:rank + 1])
            if rank == 0:
                t_render = timed(render)
                render()
                t_compact = timed(compact)
            else:
                render(); compact()
        torch.cuda.synchronize()
        nmax = max(int(c) for c in counts.tolist())
        t_place = 0.0
        if with_place:
            gathered = entries[:, :max(nmax, 1)].contiguous()
            out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
            table = {}
            def place():
                table["t"] = r.place_tiles(gathered, counts, w, h, tr, out, table.get("t"))
            t_place = timed(place)
            del gathered, out
        del slab, entries
        torch.cuda.empty_cache()
        return t_render, t_compact, nmax * TILE_ENTRY_BYTES / 1e6, t_place

    for N in (1, 2, 4, 8):
        # bench.py's pattern of layouts, one letter per batch: A = all N ranks render, B = ranks 1 .. N - 1 render and the root
        # gathers and places only
        pattern = PATTERNS.get(N, "B") if PATTERNS else ("A" if N < 3 else "B")
        pc = {c: pieces(N if c == "A" else N - 1, N > 1) for c in sorted(set(pattern))}
        nA, nB, n = pattern.count("A"), pattern.count("B"), len(pattern)
        ex = N > 1                                            # (a single GPU renders into the frames: no exchange)
        work = sum(pc[c][0] + (pc[c][1] if ex else 0.0) for c in pattern) / n                 # a rank that renders every batch
        root = (sum(pc["A"][0] + (pc["A"][1] if ex else 0.0) for _ in range(nA)) + sum(pc[c][3] for c in pattern)) / n   # placement behind its renders
        link = sum(pc[c][2] for c in pattern) / n / 1e3 / LINK_GBS * 1e3 if ex else 0.0      # (a peer's link, averaged over the pattern)
        period = max(work, link, root)
        bound = "render" if period == work else "link" if period == link else "root (its renders + every placement)"
        fs = F / period * 1e3
        if base is None: base = fs
        def ab(i, fmt="{:.2f}"):
            return " / ".join(fmt.format(pc[c][i]) if c in pc else "-" for c in "AB")
        print(f"| {N} | {pattern} | {ab(0)} | {ab(1)} | {ab(2, '{:.1f}')} | {link:.2f} | {max(pc[c][3] for c in pc):.2f} | {work:.2f} | {root:.2f} | "
              f"{bound} | {fs:.0f} | {fs / base:.2f} |")
    print()
