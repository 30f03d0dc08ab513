"""GPU box: wall time of the drop-in render(width, height) call (scene upload + kernels + copy to host)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd
from raytracer_amd.render import Renderer, _renderer
from raytracer_amd.scenes import pack_scene, generate_scene
for name, sc in (("default", None), ("64 spheres", generate_scene(64, 2)), ("1024 spheres", generate_scene(1024, 3))):
    for (w, h) in ((640, 480), (1920, 1080)):
        raytracer_amd.render(w, h, scene=sc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): img = raytracer_amd.render(w, h, scene=sc)
        t1 = time.perf_counter()
        r = _renderer()
        t2 = time.perf_counter()
        for _ in range(10): p = r.upload_scene(sc)
        t3 = time.perf_counter()
        for _ in range(10): g = r.render(w, h); torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(10): x = g.cpu()
        t5 = time.perf_counter()
        print(f"{name:12s} {w}x{h}: render() {1e2*(t1-t0):.3f} ms = upload {1e2*(t3-t2):.3f} + kernels {1e2*(t4-t3):.3f} + copy {1e2*(t5-t4):.3f}")
