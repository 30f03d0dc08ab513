import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def golden_scene_arrays(g):
    """Scene arrays of a fixture, or the float32-rounded default scene for the C1 fixtures (which
    were rendered from the reference's own unrounded module-global scene)."""
    if "scene_center" in g:
        return {k[6:]: g[k] for k in g if k.startswith("scene_")}
    from raytracer_amd.scenes import default_scene, pack_scene
    p = pack_scene(default_scene())
    return dict(center=p.geom4[:, :3], radius=p.radius, color=p.mat4[:, :3], specular=p.spec,
                reflection=p.mat4[:, 3], lights=p.lights, area_samples=np.int64(p.area_samples))


def scene_dict_from_arrays(a):
    from raytracer_amd.scenes import scene_from_arrays
    return scene_from_arrays(a["center"], a["radius"], a["color"], a["specular"], a["reflection"],
                             a["lights"], int(a["area_samples"]), refraction=a.get("refraction"),
                             refractive_index=a.get("refractive_index"), kind=a.get("kind"),
                             box_max=a.get("box_max"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.lib()
    return cpu_oracle


@pytest.fixture(scope="session")
def renderer():
    """One rt_ctx on cuda:0 through the C ABI; fails (not skips) if the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from raytracer_amd.render import Renderer
    r = Renderer(0)
    yield r
    r.close()
