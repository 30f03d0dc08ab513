"""GPU box: randomized check that none of the culls and tables changes a byte.  Random scenes (sphere count, light
positions and sizes — also lights next to or inside spheres —, cameras); each is rendered by the default
path, with the occluder table off, with the direction table off, and by the
one-thread-per-pixel kernel (linear primary scan, plain reflection scan, same shading): frames and hit maps must be
identical."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import AreaLight, Camera, Light, Vector3, generate_scene, orbit_camera

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1    # render this case alone (the others still draw their random numbers)
r = Renderer(0)
bad = 0
for case in range(n_cases):
    n = int(rng.choice([3, 8, 20, 64, 150, 400, 1000, 2040]))
    sc = generate_scene(n, int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    sph = sc["spheres"]
    if rng.integers(0, 4) == 0:      # a huge "floor" and a large far sphere: the grid's `big` list, table rows of all ones
        from raytracer_amd.scenes import Material, Sphere
        sph.append(Sphere(Vector3(0, -1003.5, -10), 1000, Material(Vector3(0.8, 0.8, 0.8), specular=0.3, reflection=0.4)))
        sph.insert(0, Sphere(Vector3(25, 5, -45), 20, Material(Vector3(0.9, 0.5, 0.5), specular=0.3, reflection=0.6)))
    def rand_pos():
        mode = rng.integers(0, 4)
        if mode == 0:   # somewhere around the scene
            return Vector3(*(rng.uniform(-15, 15), rng.uniform(-10, 15), rng.uniform(-28, 12)))
        s = sph[int(rng.integers(0, len(sph)))]
        c = np.array([s.center.x, s.center.y, s.center.z]); d = rng.normal(size=3); d /= np.linalg.norm(d)
        k = {1: 1.05, 2: 0.5, 3: 3.0}[int(mode)]     # just outside, inside, a few radii away from a sphere
        return Vector3(*(c + d * s.radius * k))
    sc["global_light"] = Light(rand_pos(), Vector3(1, 1, 1), 0.9)
    ext = float(rng.choice([0.05, 0.5, 1.0, 3.0, 8.0]))
    u = rng.normal(size=3); v = rng.normal(size=3)
    sc["area_light"] = AreaLight(rand_pos(), Vector3(*(u / np.linalg.norm(u) * ext)), Vector3(*(v / np.linalg.norm(v) * ext * rng.uniform(0.2, 1))),
                                 Vector3(1, 1, 1), 0.9, samples=int(rng.choice([1, 4, 12, 20])))
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h, D = int(rng.integers(100, 700)), int(rng.integers(60, 400)), int(rng.integers(0, 7))
    if n >= 1000: w, h = min(w, 240), min(h, 135)
    if only >= 0 and case != only:
        continue
    outs = []
    # default path / no occluder table / no direction table / neither / pixel kernel
    for table, kernel, refl in ((True, "stream", True), (False, "stream", True), (True, "stream", False), (False, "stream", False),
                                (True, "pixel", True)):
        r.set_occluder_table(table); r.set_kernel(kernel); r.set_reflect_table(refl); r.upload_scene(sc)
        outs.append(r.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True))
    r.set_kernel("stream"); r.set_occluder_table(True); r.set_reflect_table(True)
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    if not same:
        ref = outs[-1]
        for i, o in enumerate(outs[:-1]):
            dh = (o[1] != ref[1]); dc = (o[0] != ref[0]).any(-1)
            ys, xs = torch.nonzero(dh | dc, as_tuple=True)
            print(f"   config {i} vs pixel kernel: {int(dh.sum())} hit indices, {int(dc.sum())} pixels differ; first at", [(int(x), int(y)) for x, y in zip(xs[:6], ys[:6])],
                  [(int(o[1][y, x]), int(ref[1][y, x])) for x, y in zip(xs[:6], ys[:6])])
        print("   camera", None if cam is None else (cam.origin, cam.forward), "lights", sc["global_light"].position.as_tuple() if hasattr(sc["global_light"].position, "as_tuple") else sc["global_light"].position)
    lit = float((outs[0][0].max(-1).values > 0).float().mean())
    print(f"case {case}: n={n} {w}x{h} D={D} ext={ext} ns={sc['area_light'].samples} lit={lit:.3f} {'ok' if same else 'DIFFERENT'}", flush=True)
    bad += not same
print("fuzz: all identical" if bad == 0 else f"fuzz: {bad} cases differ")
sys.exit(1 if bad else 0)
