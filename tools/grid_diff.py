"""GPU box: frames with the sphere grid on vs off; counts differing pixels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import generate_scene, orbit_camera
r = Renderer(0)
from raytracer_amd.scenes import Material, Sphere, Vector3
big = generate_scene(300, 11)
big["spheres"].append(Sphere(Vector3(0, -1004, -10), 1000, Material(Vector3(0.8, 0.8, 0.8), specular=0.3, reflection=0.4)))
big["spheres"].insert(5, Sphere(Vector3(30, 0, -40), 25, Material(Vector3(0.9, 0.5, 0.5), specular=0.3, reflection=0.6)))
cases = [("128", generate_scene(128, 4), 640, 360, 5), ("200", generate_scene(200, 5), 640, 360, 8), ("1024", generate_scene(1024, 3), 960, 540, 5),
         ("2000", generate_scene(2000, 6), 480, 270, 3), ("big", big, 640, 360, 6)]
for name, sc, w, h, D in cases:
    for cname, cam in (("def", None), ("o37", orbit_camera(37, 256)), ("o150", orbit_camera(150, 256, radius=14.0))):
        out = []
        for on in (True, False):
            r.set_sphere_grid(on); r.upload_scene(sc)
            out.append(r.render(w, h, 1, max_depth=D, rng_mode="K", seed=3, camera=cam, return_hit_index=True))
        d = (out[0][0].int() - out[1][0].int()).abs().max(-1).values
        print(f"{name} {cname} D={D}: {int((d > 0).sum())} pixels differ (max {int(d.max())}), hit maps differ at {int((out[0][1] != out[1][1]).sum())}; of {w*h}", flush=True)
r.set_sphere_grid(True)
