"""GPU box, under rocprofv3 --kernel-trace: the visibility pass against tile stride and rows per rank (diagnostic)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from raytracer_amd import tiling
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera
r = Renderer(0)
w, _, n, _, _, D, desc = bench.WORKLOADS["c2"]
F, tr = 256, 8
r.upload_scene(bench.build_scene("c2"))
cams = r.pack_cameras([Camera()] * F)
cfgs = [(G, 192 * G) for G in range(1, 9)] + [(4, 4 * 48 * gx) for gx in (3, 4, 5, 6, 7, 8)] + [(7, 7 * 48 * gx) for gx in (3, 4, 5)]
for G, h in cfgs:
    rows_max = tiling.max_local_rows(h, tr, G)
    slab = torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=0, tile_stride=G)
    torch.cuda.synchronize()
    print("cfg", G, h, rows_max, flush=True)
