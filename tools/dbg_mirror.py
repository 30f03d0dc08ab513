import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import *
from oracle import cpu_oracle as co
m = Material(Vector3(1,1,1), specular=1.0, reflection=0.5)
sc = default_scene(); sc["spheres"] = [Sphere(Vector3(0,0,-5),1,m), Sphere(Vector3(0,0,5),3.9,m)]
r = Renderer(0); r.upload_scene(sc)
for kern in ("stream","pixel"):
    r.set_kernel(kern)
    for D in (0,1,2,3,4,8):
        r.reset_ray_counts()
        img = r.render(3,3,1,max_depth=D,rng_mode="C").cpu().numpy()
        o = co.render_scene(sc,3,3,max_depth=D,mode="C")
        rc = r.ray_counts()
        print(kern, D, rc, o["trace_levels"], o["hit_levels"], img[1,1], o["rgb8"][4])
