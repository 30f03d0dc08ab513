#!/usr/bin/env python3
"""Scene upload time (host-side table builds included): python tools/upload_time.py [n_spheres ...]  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import generate_scene
r = Renderer(0)
for n in [int(a) for a in sys.argv[1:]] or [8, 64, 255, 1024, 2000]:
    sc = generate_scene(n, 3)
    for on in (True, False):
        r.set_reflect_table(on)
        t0 = time.perf_counter(); r.upload_scene(sc, force=True); r.synchronize() if hasattr(r, "synchronize") else None
        print(f"{n:5d} spheres  reflect_table={on!s:5s}  upload {1e3 * (time.perf_counter() - t0):8.1f} ms", flush=True)
