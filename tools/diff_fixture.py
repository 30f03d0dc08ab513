"""Debug aid: render a golden fixture on the GPU and list the pixels that differ by more than 1/255."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import golden_scene_arrays, load_golden, scene_dict_from_arrays
from raytracer_amd.render import Renderer

name = sys.argv[1]
g = load_golden(name); m = g["meta"]
r = Renderer(0)
r.upload_scene(scene_dict_from_arrays(golden_scene_arrays(g)))
img, hit = r.render(m["width"], m["height"], m["samples"], max_depth=m["max_depth"], rng_mode=m["mode"], seed=m["seed"], return_hit_index=True)
img = img.cpu().numpy().astype(int); hit = hit.cpu().numpy()
d = np.abs(img - g["rgb8"].astype(int)).max(-1)
ys, xs = np.nonzero(d > 1)
print("bad", len(ys), "hit mismatch", int((hit != g["hit"]).sum()))
for y, x in list(zip(ys, xs))[:60]:
    print(x, y, "hit", hit[y, x], g["hit"][y, x], "gpu", img[y, x], "ref", g["rgb8"][y, x], "levels", g["trace_levels"][y, x], g["hit_levels"][y, x],
          "col", np.round(g["color"][y, x], 4))
