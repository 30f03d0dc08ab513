import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera, generate_scene, pack_scene
import numpy as np
r = Renderer(0)
def timeit(scene, n=10, F=16, D=5, w=1920, h=1080):
    r.upload_scene(scene)
    out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda")
    cams=[Camera()]*F
    for _ in range(2): r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=out)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=out)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for nsph in (8, 64):
    sc = generate_scene(nsph, 2 if nsph==64 else 1, keep_default=(nsph==8))
    t_full = timeit(sc)
    t_d0 = timeit(sc, D=0)
    # same spheres moved behind the camera: every primary misses (fill only)
    p = pack_scene(sc)
    for s in sc["spheres"]: s.center.z = abs(s.center.z) + 30
    t_miss = timeit(sc)
    print(f"N={nsph}: full D=5 {t_full:.3f} ms, D=0 {t_d0:.3f} ms, all-miss (fill only) {t_miss:.3f} ms per 16 frames")
