"""GPU box, ONE GPU: the pieces of an N-GPU row-tile step measured one by one, and the scaling curve they predict
(DESIGN.md 7).  For each workload and N in 1, 2, 4, 8: rank 0's render share of a batch (tile_first = 0, tile_stride = N),
rt_compact_tiles of its slab, the bytes a rank hands to the gather, and rt_place_tiles of all N ranks' entries on the root
(every rank's share is rendered here, one after another, to have real entries).  Model of the steady state of the
pipelined exchange (raytracer_amd/distributed.py): a batch leaves every
    period(N) = max( render + compact [a rank that renders],  bytes per rank / link rate [each peer has its own xGMI link into
                     GPU 0],  root: render + compact + place [N = 2: the placement runs behind the root's own render], place alone
                     [N >= 3: the root does not render; N - 1 ranks share the row tiles] )
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
workloads = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--frames"]
for wl in (workloads or ["c2", "t64", "c4"]):
    w, h, n, _, _, D, desc = bench.WORKLOADS[wl]
    F = max(1, FRAMES // 16) if wl == "c4" else FRAMES
    r.upload_scene(bench.build_scene(wl))
    cams = r.pack_cameras([Camera()] * F)
    print(f"## {wl}: {desc}; batches of {F} frames, row tiles of {tr} rows")
    print("| N | ranks that render | render share ms | compact ms | occupied tiles of the first renderer | MB per rank | link ms @153 GB/s | "
          "place ms (root) | renderers' period ms | root's period ms | bound | frames/s | speed-up |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    base = None
    for N in (1, 2, 4, 8):
        G = N if N < 3 else N - 1          # bench.py: from three GPUs on the root gathers and places, it does not render
        rows_max = tiling.max_local_rows(h, tr, G)
        cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
        slabs = [torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0") for _ in range(1)]
        entries = torch.empty((G, cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
        counts = torch.zeros(G, dtype=torch.int32, device="cuda:0")
        t_render = t_compact = None
        for rank in range(G):
            def render():
                r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slabs[0], tile_rows=tr, tile_first=rank, tile_stride=G)
            def compact():
                r.compact_tiles(slabs[0], entries[rank], counts[rank:rank + 1])
            if rank == 0:
                t_render = timed(render)
                render()
                t_compact = timed(compact)
            else:
                render(); compact()
        torch.cuda.synchronize()
        cnt = [int(c) for c in counts.tolist()]
        nmax = max(cnt)
        gathered = entries[:, :max(nmax, 1)].contiguous()
        out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
        table = {}
        def place():
            table["t"] = r.place_tiles(gathered, counts, w, h, tr, out, table.get("t"))
        t_place = timed(place) if N > 1 else 0.0
        mb = nmax * TILE_ENTRY_BYTES / 1e6
        t_link = mb / 1e3 / LINK_GBS * 1e3 if N > 1 else 0.0
        work = t_render + (t_compact if N > 1 else 0.0)      # (a single GPU renders into the frames: no exchange)
        root = work + t_place if G == N else t_place          # (placement behind the root's own render: tools/root_emulation.py)
        period = max(work, t_link, root)
        bound = "link" if period == t_link else "root (render + placement)" if (period == root and G == N and N > 1) else \
                "root (placement)" if period == root and N > 1 else "render"
        fs = F / period * 1e3
        if base is None: base = fs
        print(f"| {N} | {G} | {t_render:.2f} | {t_compact:.2f} | {cnt[0]} | {mb:.1f} | {t_link:.2f} | {t_place:.2f} | {work:.2f} | {root:.2f} | "
              f"{bound} | {fs:.0f} | {fs / base:.2f} |")
        del slabs, entries, gathered, out
        torch.cuda.empty_cache()
    print()
