"""GPU box: random scenes with cubes (opaque and refractive, boxes around lights, around the camera, behind surfaces),
one digest per case of frame + hit map, rendered three times: default, direction table off, occluder table off (must agree).
Run once with the shipped library and once with a -DRT_CUBE_SHADOW_CULL=0 build (RT355_LIB=...), and compare the two outputs:
the candidate-box cull of lighting() must not change a byte either.
    python tools/fuzz_cubes.py [cases] [seed] > a.txt;  RT355_LIB=$PWD/nocull.so python tools/fuzz_cubes.py [cases] [seed] > b.txt;  cmp a.txt b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import AreaLight, Cube, Light, Material, Vector3, generate_scene, orbit_camera

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = Renderer(0)
for case in range(n_cases):
    n = int(rng.choice([0, 3, 8, 20, 64]))
    sc = generate_scene(max(n, 1), int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    if n == 0:
        sc["spheres"] = []
    objs = list(sc["spheres"])
    glass = rng.integers(0, 2) == 1
    for s_ in objs:
        if glass and rng.random() < 0.3:
            s_.material.refraction = float(rng.uniform(0.2, 0.9)); s_.material.refractive_index = float(rng.choice([0.8, 1.33, 1.5]))
    def rand_pos():
        return np.array([rng.uniform(-8, 8), rng.uniform(-4, 8), rng.uniform(-16, 8)])
    lp, ap = rand_pos(), rand_pos()
    for _ in range(int(rng.choice([1, 2, 5, 12, 40]))):
        kind = int(rng.integers(0, 5))
        h = rng.uniform(0.1, 1.5, size=3)
        if kind == 0 or not objs:      # anywhere
            c = rand_pos()
        elif kind == 1:                # around or next to a light
            c = (lp if rng.integers(0, 2) else ap) + rng.normal(size=3) * 0.7
        elif kind == 2:                # around the camera
            c = rng.normal(size=3) * 0.5; h = rng.uniform(0.5, 3.0, size=3)
        else:                          # on the line from a sphere to a light, on either side and beyond
            balls = [o for o in objs if hasattr(o, "center")]
            s_ = balls[int(rng.integers(0, len(balls)))] if balls else None
            sc_ = np.array([s_.center.x, s_.center.y, s_.center.z]) if s_ is not None else rand_pos()
            c = sc_ + rng.uniform(-1.5, 2.0) * ((lp if rng.integers(0, 2) else ap) - sc_)
        m = Material(Vector3(*rng.uniform(0.1, 1.0, size=3)), specular=float(rng.uniform(0, 1)), reflection=float(rng.choice([0.0, 0.4])))
        if glass and rng.random() < 0.3:
            m.refraction = float(rng.uniform(0.2, 0.9)); m.refractive_index = float(rng.choice([1.0, 1.33, 1.5]))
        objs.insert(int(rng.integers(0, len(objs) + 1)), Cube(Vector3(*(c - h)), Vector3(*(c + h)), m))
    sc["spheres"] = objs
    sc["global_light"] = Light(Vector3(*lp), Vector3(1, 1, 1), 0.9)
    ext = float(rng.choice([0.05, 0.5, 1.0, 3.0, 8.0]))
    u = rng.normal(size=3); v = rng.normal(size=3)
    sc["area_light"] = AreaLight(Vector3(*ap), Vector3(*(u / np.linalg.norm(u) * ext)), Vector3(*(v / np.linalg.norm(v) * ext * rng.uniform(0.2, 1))),
                                 Vector3(1, 1, 1), 0.9, samples=int(rng.choice([1, 4, 12, 20])))
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h_, D = int(rng.integers(100, 500)), int(rng.integers(60, 300)), int(rng.integers(0, 6))
    dig = []
    # (the one-thread-per-pixel kernel takes no cubes.)  In-process: the default path, then without the direction table of
    # secondary rays (boxes as sources and targets), then without the occluder table — the same bytes (glass: integer sums)
    for refl, occl in ((True, True), (False, True), (True, False)):
        r.set_reflect_table(refl); r.set_occluder_table(occl); r.upload_scene(sc)
        img, hit = r.render(w, h_, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
        hsh = hashlib.sha256(img.cpu().numpy().tobytes() + hit.cpu().numpy().tobytes()).hexdigest()[:16]
        dig.append(hsh)
        lit = float((img.max(-1).values > 0).float().mean())
    r.set_reflect_table(True); r.set_occluder_table(True)
    same = "same" if len(set(dig)) == 1 else "DIFFERENT " + " ".join(dig)
    print(f"case {case}: spheres={n} objects={len(objs)} glass={bool(glass)} {w}x{h_} D={D} ext={ext} lit={lit:.3f} digest={dig[0]} tables on/off: {same}", flush=True)
