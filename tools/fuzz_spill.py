#!/usr/bin/env python3
"""GPU-only fuzz of the refractive work queue: random scenes of mostly glass spheres and cubes, depth up to 12,
rendered with the spill stack off / automatic / capped at 2 and 5 blocks.  Same tree nodes every way: hit maps
and ray counts must be identical, images equal up to float summation order.   python tools/fuzz_spill.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd.render import Renderer  # noqa: E402
from raytracer_amd.scenes import Cube, Material, Vector3, generate_scene, orbit_camera  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = Renderer(0)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([3, 6, 12, 24]))
    sc = generate_scene(n, int(rng.integers(1, 1000)))
    objs = list(sc["spheres"])
    for s_ in objs:
        if rng.random() < 0.8:
            s_.material.refraction = float(rng.uniform(0.3, 0.9))
            s_.material.refractive_index = float(rng.choice([1.0, 1.0, 1.33, 1.5]))
            s_.material.reflection = float(rng.choice([0.3, 0.6]))
    for _ in range(int(rng.integers(0, 5))):
        c = np.array([rng.uniform(-4, 4), rng.uniform(-2.5, 2.5), rng.uniform(-14, -3)])
        e = rng.uniform(0.3, 1.5, 3)
        m = Material(Vector3(*rng.uniform(0.3, 1, 3)), specular=0.3, reflection=0.5, refraction=0.5, refractive_index=1.0)
        objs.insert(int(rng.integers(0, len(objs) + 1)), Cube(Vector3(*(c - e)), Vector3(*(c + e)), m))
    sc["spheres"] = objs
    cam = None if case % 2 == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(4, 9)))
    D = int(rng.choice([3, 5, 8, 12]))
    w, h = (320, 180) if D <= 5 else (128, 72)
    out = []
    for on in (False, True, 2, 5):
        r.set_queue_spill(on)
        r.upload_scene(sc)
        r.reset_ray_counts()
        img, hit = r.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
        out.append((img.to(torch.int16), hit, r.ray_counts()))
    a, ha, ca = out[0]
    ok = True
    for b, hb, cb in out[1:]:
        d = (a - b).abs()
        ok = ok and torch.equal(ha, hb) and ca == cb and int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.02
    bad += not ok
    print(f"case {case}: {n} spheres, {len(objs) - n} cubes, D {D}, rays/pixel {ca['total'] / (w * h):.1f} {'ok' if ok else 'MISMATCH'}", flush=True)
r.set_queue_spill(True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
