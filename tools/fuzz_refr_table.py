"""GPU box: refractive scenes with the direction table on and off — the tree must be the same (ray counts, hit maps),
frames equal up to float summation order.  python tools/fuzz_refr_table.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import generate_scene, orbit_camera

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = Renderer(0)
bad = 0
for case in range(n_cases):
    n = int(rng.choice([4, 8, 20, 64, 150, 250]))
    sc = generate_scene(n, int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    frac = float(rng.choice([0.1, 0.35, 0.8]))
    for s_ in sc["spheres"]:
        if rng.random() < frac:
            s_.material.refraction = float(rng.uniform(0.2, 0.9))
            s_.material.refractive_index = float(rng.choice([0.8, 1.0, 1.33, 1.5, 2.2]))
    if not any(s_.material.refraction > 0 for s_ in sc["spheres"]):
        sc["spheres"][0].material.refraction = 0.5
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h, D = int(rng.integers(100, 400)), int(rng.integers(60, 250)), int(rng.integers(1, 8))
    outs = []
    for on in (True, False):
        r.set_reflect_table(on); r.upload_scene(sc); r.reset_ray_counts()
        img, hit = r.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
        outs.append((img.to(torch.int16), hit, r.ray_counts()))
    r.set_reflect_table(True)
    (a, ha, ca), (b, hb, cb) = outs
    d = (a - b).abs()
    same = ca == cb and torch.equal(ha, hb) and int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    print(f"case {case}: n={n} glass {frac} {w}x{h} D={D} rays {ca['total']} {'ok' if same else 'DIFFERENT ' + str(ca) + str(cb)}", flush=True)
    bad += not same
print("fuzz: all trees identical" if bad == 0 else f"fuzz: {bad} cases differ")
sys.exit(1 if bad else 0)
