"""Edge semantics of the reference path, exercised on the oracle with hand-made scenes
(SURVEY.md §2.1, §4 unit rows).  The same scenes run on the GPU in test_gpu_parity.py."""
import numpy as np
import pytest

from raytracer_amd.scenes import (AreaLight, Light, Material, Sphere, Vector3, default_scene,
                                  pack_scene)


def make(spheres, ns=12):
    s = default_scene()
    s["spheres"] = spheres
    s["area_light"].samples = ns
    return s


def mat(c=(1, 1, 1), spec=1.0, refl=0.0):
    return Material(Vector3(*c), specular=spec, reflection=refl)


def centre(oracle, scene, w=3, h=3, **kw):
    """Pixel (1,1) of a 3x3 frame in mode C looks straight down -z."""
    kw.setdefault("mode", "C")
    return oracle.render_scene(scene, w, h, pixels=[[1, 1]], **kw)


def test_empty_scene_is_black(oracle):
    o = oracle.render_scene(make([]), 8, 6)
    assert not o["rgb8"].any() and (o["hit"] == -1).all() and (o["trace_levels"] == 1).all()


def test_tie_goes_to_lowest_index(oracle):
    a = Sphere(Vector3(0, 0, -5), 1, mat((1, 0, 0)))
    b = Sphere(Vector3(0, 0, -5), 1, mat((0, 1, 0)))
    assert centre(oracle, make([a, b]))["hit"][0] == 0      # strict `<` (raytracer.py:162)
    assert centre(oracle, make([b, a]))["hit"][0] == 0


def test_origin_inside_sphere_misses(oracle):
    big = Sphere(Vector3(0, 0, 0), 2, mat())                 # near root negative, far root never tried
    o = centre(oracle, make([big]))
    assert o["hit"][0] == -1 and not o["rgb8"].any()


def test_sphere_behind_camera_misses(oracle):
    assert centre(oracle, make([Sphere(Vector3(0, 0, 5), 1, mat())]))["hit"][0] == -1


def test_nearest_of_two(oracle):
    far = Sphere(Vector3(0, 0, -9), 1, mat())
    near = Sphere(Vector3(0, 0, -4), 1, mat())
    assert centre(oracle, make([far, near]))["hit"][0] == 1


def test_depth_limit_counts(oracle):
    """Two facing mirrors: every level hits, so exactly D+1 levels are traced (raytracer.py:233)."""
    m = mat(refl=0.5)
    sc = make([Sphere(Vector3(0, 0, -5), 1, m), Sphere(Vector3(0, 0, 5), 3.9, m)])
    for D in (0, 1, 3, 5, 8):
        o = centre(oracle, sc, max_depth=D)
        assert o["trace_levels"][0] == D + 1 and o["hit_levels"][0] == D + 1


def test_no_reflection_stops_recursion(oracle):
    sc = make([Sphere(Vector3(0, 0, -5), 1, mat(refl=0.0))])
    o = centre(oracle, sc)
    assert o["trace_levels"][0] == 1 and o["hit_levels"][0] == 1


def test_quantisation_truncates(oracle):
    sc = make([Sphere(Vector3(0, 0, -5), 1, mat((1, 1, 1), spec=0.0))])
    o = centre(oracle, sc)
    c = np.clip(o["color"][0], 0, 1)
    assert np.array_equal(o["rgb8"][0], np.floor(c * 255).astype(np.uint8))


def test_area_sample_count_matters_only_through_mean(oracle):
    """Mode C puts every area sample at the light centre, so the result is independent of samples."""
    base = centre(oracle, make(default_scene()["spheres"], ns=12))["color"]
    for ns in (1, 4, 16):
        c = centre(oracle, make(default_scene()["spheres"], ns=ns))["color"]
        assert np.allclose(c, base, rtol=1e-12)


def test_pack_scene_rejects_out_of_scope():
    s = default_scene()
    s["spheres"][0].material.refraction = 0.5
    s["spheres"][0].material.refractive_index = 1.5
    pk = pack_scene(s)                      # refraction is in scope
    assert pk.has_refraction and pk.refr[0].tolist() == [0.5, 1.5] and pk.refr[1].tolist() == [0.0, 1.0]
    s["spheres"][0].material.refractive_index = 0.0
    with pytest.raises(ValueError):
        pack_scene(s)
    s = default_scene()
    s["spheres"].append(object())
    with pytest.raises(TypeError):
        pack_scene(s)
    s = default_scene()
    del s["area_light"]
    with pytest.raises(KeyError):
        pack_scene(s)
