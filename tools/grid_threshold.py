"""GPU box: batch render time with the sphere grid on and off for several sphere counts (to place kGridFrom)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera, generate_scene
r = Renderer(0)
F, w, h = 32, 1920, 1080
cams = r.pack_cameras([Camera()] * F)
out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
for n in (128, 160, 200, 256, 384, 512, 1024, 2048):
    sc = generate_scene(n, 5)
    res = []
    for on in (True, False):
        r.set_sphere_grid(on); r.upload_scene(sc)
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r.render_frames(cams, range(it * F, it * F + F), w, h, 1, max_depth=5, out=out); b.record()
            torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res.append(min(ts[1:]))
    print(f"n={n}: grid {res[0]:.2f} ms, plain scan {res[1]:.2f} ms per {F} frames  ({res[1] / res[0]:.2f}x)", flush=True)
r.set_sphere_grid(True)
