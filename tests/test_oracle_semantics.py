"""Edge semantics of the reference path, exercised on the oracle with hand-made scenes
(SURVEY.md §2.1, §4 unit rows).  The same scenes run on the GPU in test_gpu_parity.py."""
import numpy as np
import pytest

from raytracer_amd.scenes import (AreaLight, Cube, Light, Material, Sphere, Vector3, cube_scene, default_scene,
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


# ---- cubes (raytracer.py:106-150, :157-169) -------------------------------------
def glass(c=(1, 1, 1)):
    return Material(Vector3(*c), specular=0.0, reflection=0.0, refraction=0.5, refractive_index=1.0)


def test_cube_in_front_hits_near_face_and_shadows_itself(oracle):
    """The centre ray (0, 0, -1) meets the box at z = -4 (t = 4).  Its shadow rays start 0.001 outside that
    face and point away from the box, yet intersect_cube returns t_max = -0.001 / d_z for them (:137) and
    `if t and t < min_distance` (:162) accepts it: an opaque cube is black."""
    box = Cube(Vector3(-1, -1, -6), Vector3(1, 1, -4), mat((1, 1, 1), spec=0.0))
    o = centre(oracle, make([box]))
    assert o["hit"][0] == 0 and o["hit_levels"][0] == 1 and not o["color"].any()
    # the same box with a refractive material is lit: its nearest (t < 0) shadow hit is refractive (:187, :202)
    lit = centre(oracle, make([Cube(box.min_point, box.max_point, glass())]))
    assert lit["hit"][0] == 0 and lit["color"][0].min() > 0


def test_cube_behind_camera_is_hit_with_negative_t_and_beats_spheres(oracle):
    box = Cube(Vector3(-1, -1, 4), Vector3(1, 1, 6), glass())
    sph = Sphere(Vector3(0, 0, -5), 1, mat())
    assert centre(oracle, make([sph, box]))["hit"][0] == 1      # t = t_max = -4 < the sphere's 4
    assert centre(oracle, make([box, sph]))["hit"][0] == 0
    # off the box's line the (larger) sphere is seen again
    o = oracle.render_scene(make([Sphere(Vector3(0, 0, -5), 2.5, mat()), box]), 64, 64, mode="C")
    assert set(np.unique(o["hit"])) == {-1, 0, 1}


def test_cube_camera_inside_returns_far_face(oracle):
    """t_min < 0 -> t_max (:137): from inside, the far face at z = -9 is hit, normal (0, 0, -1) (:148)."""
    box = Cube(Vector3(-3, -2, -9), Vector3(3, 2, 1), glass((0.5, 0.25, 1.0)))
    o = centre(oracle, make([box]), max_depth=0)
    assert o["hit"][0] == 0
    # colour = m (1 - f) * light at P = (0, 0, -9), N = (0, 0, -1): both lights are behind the face (N.L < 0)
    assert not o["color"].any()


def test_cube_parallel_ray_outside_slab_misses(oracle):
    """|d_x|, |d_y| < 1e-8 for the centre ray: the box is missed iff the origin lies outside its x / y slabs (:124-126)."""
    assert centre(oracle, make([Cube(Vector3(0.5, -1, -6), Vector3(1.5, 1, -4), glass())]))["hit"][0] == -1
    assert centre(oracle, make([Cube(Vector3(-0.5, -1, -6), Vector3(1.5, 1, -4), glass())]))["hit"][0] == 0


def test_cube_normal_face_order(oracle):
    """cube_normal tests x-min, x-max, y-min, ... in that order (:144-149): at a point on two faces the
    earlier one wins.  A glass box lit from the point light only through the face normal: at the top
    edge x = -1 the normal is (-1, 0, 0), not (0, 1, 0)."""
    s = make([Cube(Vector3(-1, -1, -6), Vector3(1, 1, -4), glass())])
    o = oracle.render_scene(s, 65, 65, mode="C", max_depth=0)
    assert (o["hit"] == 0).sum() > 100


def test_pack_scene_orders_cubes_after_spheres():
    pk = pack_scene(cube_scene())
    assert pk.n_spheres == 3 and pk.n_cubes == 3 and list(pk.order) == [1, 2, 4, 0, 3, 5]
    assert pk.list_index(np.array([-1, 0, 3, 5])).tolist() == [-1, 1, 0, 5]
    o = pk.objects_in_list_order()
    assert o["kind"].tolist() == [1, 0, 0, 1, 0, 1]
    assert o["p0"][0].tolist() == pytest.approx([-1.5, -0.35, -4.3]) and o["p1"][5].tolist() == pytest.approx([-2.6, -2.4, -11.4])
    assert o["radius"].tolist() == pytest.approx([0, 1, 1.5, 0, 0.75, 0])
    assert not pk.has_refraction and pack_scene(cube_scene(glass=True)).has_refraction
