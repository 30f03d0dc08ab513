"""Debug aid (GPU box): per-rank render time of a 64-frame config-2 batch when the rows are sharded over G ranks
(what rank 0 of an N-GPU bench run launches), measured on one GPU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from raytracer_amd import tiling
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera
w, h, D, F = 1920, 1080, 5, 64
r = Renderer(0)
r.upload_scene(bench.build_scene("c2"))
cams = r.pack_cameras([Camera()] * F)
for G in (1, 2, 4, 8):
    rows = tiling.max_local_rows(h, 8, G)
    out = torch.empty((F, rows, w, 3), dtype=torch.uint8, device="cuda:0")
    ts = []
    for it in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r.render_frames(cams, range(it * F, (it + 1) * F), w, h, 1, max_depth=D, out=out, tile_rows=8, tile_first=0, tile_stride=G)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(20):
        r.render_frames(cams, range(it * F, (it + 1) * F), w, h, 1, max_depth=D, out=out, tile_rows=8, tile_first=0, tile_stride=G)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"      host time per call {1e3 * (t1 - t0) / 20:.3f} ms, wall per call incl. GPU {1e3 * (t2 - t0) / 20:.3f} ms")
    print(f"G={G}: {min(ts[2:]):.3f} ms per 64-frame batch of this rank's rows ({rows} rows)")
