"""GPU box, under rocprofv3 --kernel-trace: the launch pair against the number of frames in the batch (the fixed cost of a
launch: start-up and tail of the persistent kernel).  Diagnostic."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera
r = Renderer(0)
w, h, n, _, _, D, desc = bench.WORKLOADS["c2"]
r.upload_scene(bench.build_scene("c2"))
for F in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    cams = r.pack_cameras([Camera()] * F)
    out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=out)
    torch.cuda.synchronize()
    print("cfg", F, flush=True)
