"""Live check of the oracle against the imported reference (build container only: skipped where
/root/reference is absent, e.g. on the GPU box)."""
import numpy as np
import pytest

from oracle import ref_harness
from raytracer_amd.scenes import (Cube, Material, Vector3, cube_scene, default_scene, generate_scene, orbit_camera,
                                  refractive_scene)

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(),
                                reason="/root/reference is not present")


@pytest.mark.parametrize("mode,D,spp", [("C", 5, 1), ("K", 3, 2), ("K", 8, 1)])
def test_small_frames_bit_exact(oracle, mode, D, spp):
    sc = default_scene()
    r = ref_harness.render_full(sc, 48, 36, spp, D, mode, 21)
    o = oracle.render_scene(sc, 48, 36, samples=spp, max_depth=D, mode=mode, seed=21)
    assert np.array_equal(o["color"].reshape(36, 48, 3), r["color"])
    assert np.array_equal(o["hit"].reshape(36, 48), r["hit"])


def test_random_pixels_generated_scene_and_camera(oracle):
    sc = generate_scene(16, 9)
    cam = orbit_camera(5)
    px = np.random.default_rng(1).integers(0, [1920, 1080], (150, 2))
    r = ref_harness.render_pixels(sc, 1920, 1080, px, 1, 5, "K", 2, camera=cam)
    o = oracle.render_scene(sc, 1920, 1080, pixels=px, max_depth=5, mode="K", seed=2, camera=cam)
    assert np.array_equal(o["color"], r["color"]) and np.array_equal(o["hit"], r["hit"])


def test_keyed_rng_definition_matches(oracle):
    """Mode K's hash: Python restatement in the harness vs the C++ one, through rendered jitter."""
    sc = default_scene()
    a = ref_harness.render_pixels(sc, 64, 48, [[30, 24], [31, 24]], 1, 0, "K", 0xDEADBEEFCAFE)
    b = oracle.render_scene(sc, 64, 48, pixels=[[30, 24], [31, 24]], max_depth=0, mode="K", seed=0xDEADBEEFCAFE)
    assert np.array_equal(a["color"], b["color"])


@pytest.mark.parametrize("mode,D,spp", [("C", 5, 1), ("K", 5, 2), ("K", 2, 1)])
def test_refraction_bit_exact(oracle, mode, D, spp):
    """compute_refraction (raytracer.py:213-230, :252-254) and the refractive-occluder rule of the
    shadow test (:187, :202)."""
    sc = refractive_scene()
    r = ref_harness.render_full(sc, 64, 48, spp, D, mode, 5)
    o = oracle.render_scene(sc, 64, 48, samples=spp, max_depth=D, mode=mode, seed=5)
    assert np.array_equal(o["color"].reshape(48, 64, 3), r["color"])
    assert np.array_equal(o["hit"].reshape(48, 64), r["hit"])
    assert np.array_equal(o["trace_levels"].reshape(48, 64), r["trace_levels"])


@pytest.mark.parametrize("glass,mode,D,spp", [(False, "C", 5, 1), (False, "K", 5, 2), (True, "K", 3, 1), (True, "C", 6, 1)])
def test_cubes_bit_exact(oracle, glass, mode, D, spp):
    """Cube members of the object list: intersect_cube / cube_normal (raytracer.py:106-150) through the
    type dispatch of find_nearest_intersection (:157-169), including the accepted t < 0 (:137, :162)."""
    sc = cube_scene(glass)
    r = ref_harness.render_full(sc, 64, 48, spp, D, mode, 5)
    o = oracle.render_scene(sc, 64, 48, samples=spp, max_depth=D, mode=mode, seed=5)
    assert np.array_equal(o["color"].reshape(48, 64, 3), r["color"])
    assert np.array_equal(o["hit"].reshape(48, 64), r["hit"])
    assert np.array_equal(o["trace_levels"].reshape(48, 64), r["trace_levels"])
    assert np.array_equal(o["hit_levels"].reshape(48, 64), r["hit_levels"])


def test_cube_camera_inside_and_axis_parallel_rays(oracle):
    """The camera inside a box (t_min < 0 -> t_max, :137), a box behind the camera (t < 0 accepted, :162)
    and rays parallel to slabs (|d| < 1e-8, :124-126: the centre column / row of an odd-sized frame)."""
    m = Material(Vector3(0.9, 0.7, 0.5), specular=0.5, reflection=0.3)
    for box in (Cube(Vector3(-3, -2, -9), Vector3(3, 2, 1), m),          # camera inside
                Cube(Vector3(-0.5, -0.5, 2), Vector3(0.5, 0.5, 3), m),     # behind the camera
                Cube(Vector3(-1, -1, -6), Vector3(1, 1, -4), m)):         # in front, axis-parallel centre ray
        sc = default_scene()
        sc["spheres"] = [sc["spheres"][1], box, sc["spheres"][2]]
        r = ref_harness.render_full(sc, 33, 17, 1, 4, "C", 0)
        o = oracle.render_scene(sc, 33, 17, samples=1, max_depth=4, mode="C", seed=0)
        assert np.array_equal(o["color"].reshape(17, 33, 3), r["color"])
        assert np.array_equal(o["hit"].reshape(17, 33), r["hit"])
        assert (r["hit"] == 1).any()
