"""Debug aid (GPU box): quick parity of the library named by RT355_LIB against two golden fixtures."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import golden_scene_arrays, load_golden, scene_dict_from_arrays
from raytracer_amd.render import Renderer
r = Renderer(0)
for name in ("c1_default_640x480_D3_modeK", "c2_8sph_1920x1080_D5", "pin_default_160x120_D5_spp2_modeK", "cube_opaque_200x150_D5_spp2_modeK"):
    g = load_golden(name); m = g["meta"]
    r.upload_scene(scene_dict_from_arrays(golden_scene_arrays(g)))
    img = r.render(m["width"], m["height"], m["samples"], max_depth=m["max_depth"], rng_mode=m["mode"], seed=m["seed"]).cpu().numpy().astype(int)
    if "pixels" in g:
        px = g["pixels"]; img = img[px[:, 1], px[:, 0]]
    d = np.abs(img.reshape(-1, 3) - g["rgb8"].astype(int).reshape(-1, 3)).max(1)
    print(os.environ.get("RT355_LIB", "default"), name, "within 1/255:", (d <= 1).mean())
    assert (d <= 1).mean() >= 0.999
