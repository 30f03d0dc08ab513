"""Scene description for the rt355 render path.

Host-side mirror of the reference's scene types (``/root/reference/raytracer.py:10-87``):
``Vector3``, ``Material``, ``Sphere``, ``Cube``, ``Light``, ``AreaLight`` and the module-global
``scene`` dict with keys ``spheres`` / ``global_light`` / ``area_light`` (``spheres`` is the object
list the reference scans, raytracer.py:156; its members may be ``Cube`` objects, :157-160).  Attribute names are the
reference's, so a scene dict built for the reference can be handed to :func:`pack_scene`
unchanged (duck-typed: anything exposing ``.x/.y/.z`` etc. works).

The device layout is struct-of-arrays:

* ``geom4[N,4]`` = ``cx, cy, cz, r*r``     (float32, 16 B per sphere, staged into LDS)
* ``mat4[N,4]``  = ``R, G, B, reflection`` (float32)
* ``spec[N]``    = ``specular``            (float32)
* cubes: ``cube_box[M,6]`` = ``min xyz, max xyz`` with ``cube_mat4`` /
  ``cube_spec`` / ``cube_refr`` alike; on the device the cubes follow the spheres (object index
  ``N + j``), ``order`` maps that packed index back to the position in the reference's list

plus one ``rt_lights`` block (point light + area light) and one ``rt_camera`` block.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field

import numpy as np

__all__ = [
    "Vector3", "Material", "Sphere", "Cube", "Light", "AreaLight", "Camera",
    "default_scene", "generate_scene", "pack_scene", "scene_from_arrays",
    "orbit_camera", "cube_scene", "refractive_scene", "PackedScene", "RtLights", "RtCamera",
    "MAX_SPHERES", "MAX_AREA_SAMPLES",
]

MAX_SPHERES = 2048        # rt355.h RT_MAX_SPHERES (LDS: 32 B per sphere)
MAX_AREA_SAMPLES = 64     # rt355.h RT_MAX_AREA_SAMPLES


@dataclass
class Vector3:
    """3-vector with the reference's field names (raytracer.py:10-14). Only data here:
    the arithmetic lives in the HIP kernel (float3 register math)."""
    x: float
    y: float
    z: float

    def as_tuple(self):
        return (float(self.x), float(self.y), float(self.z))


@dataclass
class Material:
    """raytracer.py:42-48 (refraction: raytracer.py:213-230, 252-254)."""
    color: Vector3
    specular: float = 0
    reflection: float = 0
    refraction: float = 0
    refractive_index: float = 1


@dataclass
class Sphere:
    """raytracer.py:50-54."""
    center: Vector3
    radius: float
    material: Material


@dataclass
class Cube:
    """Axis-aligned box, raytracer.py:56-60 (intersect_cube :106-137, cube_normal :142-150)."""
    min_point: Vector3
    max_point: Vector3
    material: Material


@dataclass
class Light:
    """Point light, raytracer.py:62-66."""
    position: Vector3
    color: Vector3
    intensity: float


@dataclass
class AreaLight:
    """Parallelogram light, raytracer.py:68-75 (``samples`` defaults to 4 as in the reference)."""
    position: Vector3
    u: Vector3
    v: Vector3
    color: Vector3
    intensity: float
    samples: int = 4


@dataclass
class Camera:
    """Pinhole camera. The reference has none: its primary ray is
    ``Ray((0,0,0), normalize((u, v, -1)))`` (raytracer.py:295). This struct generalises that to
    ``origin + normalize(right*u + up*v + forward)`` and reduces to the reference exactly for
    the default values (multiplying by 1.0 / adding 0.0 is exact)."""
    origin: tuple = (0.0, 0.0, 0.0)
    right: tuple = (1.0, 0.0, 0.0)
    up: tuple = (0.0, 1.0, 0.0)
    forward: tuple = (0.0, 0.0, -1.0)

    def as_array(self) -> np.ndarray:
        a = np.array([self.origin, self.right, self.up, self.forward], dtype=np.float32)
        if a.shape != (4, 3) or not np.all(np.isfinite(a)):
            raise ValueError("camera must be four finite 3-vectors")
        return a


def default_scene() -> dict:
    """The reference's shipped scene, raytracer.py:78-87 (values restated, not imported)."""
    return {
        "global_light": Light(Vector3(0, 10, 10), Vector3(1, 1, 1), 0.9),
        "area_light": AreaLight(Vector3(5, 5, 5), Vector3(1, 0, 0), Vector3(0, 1, 0),
                                Vector3(1, 1, 1), 0.9, samples=12),
        "spheres": [
            Sphere(Vector3(0, 0, -5), 1,
                   Material(Vector3(1, 0.59, 0.35), specular=1.0, reflection=0.4)),
            Sphere(Vector3(-2.5, 0, -7), 1.5,
                   Material(Vector3(0.35, 1, 0.63), specular=1.0, reflection=0.6)),
            Sphere(Vector3(2.5, 0, -6), 0.75,
                   Material(Vector3(0.35, 1, 1), specular=1.0, reflection=0.7)),
        ],
    }


# --------------------------------------------------------------------------------------
# ctypes mirrors of include/rt355.h
# --------------------------------------------------------------------------------------
class RtCamera(ctypes.Structure):
    _fields_ = [("origin", ctypes.c_float * 3), ("right", ctypes.c_float * 3),
                ("up", ctypes.c_float * 3), ("forward", ctypes.c_float * 3)]

    @classmethod
    def from_camera(cls, cam: "Camera | None") -> "RtCamera":
        a = (cam or Camera()).as_array()
        out = cls()
        for i, name in enumerate(("origin", "right", "up", "forward")):
            getattr(out, name)[:] = [float(v) for v in a[i]]
        return out


class RtLights(ctypes.Structure):
    _fields_ = [("point_pos", ctypes.c_float * 3), ("point_color", ctypes.c_float * 3),
                ("point_intensity", ctypes.c_float),
                ("area_pos", ctypes.c_float * 3), ("area_u", ctypes.c_float * 3),
                ("area_v", ctypes.c_float * 3), ("area_color", ctypes.c_float * 3),
                ("area_intensity", ctypes.c_float), ("area_samples", ctypes.c_int32)]


@dataclass
class PackedScene:
    geom4: np.ndarray        # [N,4] float32: cx cy cz r^2
    mat4: np.ndarray         # [N,4] float32: R G B reflection
    spec: np.ndarray         # [N]   float32
    radius: np.ndarray       # [N]   float32 (kept for round-tripping to objects)
    lights: np.ndarray       # [20]  float32: ppos3 pcol3 pint | apos3 au3 av3 acol3 aint
    area_samples: int
    refr: np.ndarray = None  # [N,2] float32: refraction, refractive_index (Material defaults 0, 1)
    cube_box: np.ndarray = None   # [M,6] float32: min xyz, max xyz
    cube_mat4: np.ndarray = None  # [M,4]
    cube_spec: np.ndarray = None  # [M]
    cube_refr: np.ndarray = None  # [M,2]
    order: np.ndarray = None      # [N+M] int32: position in the reference's object list of packed object k
    _ct: RtLights = field(default=None, repr=False)

    def __post_init__(self):
        if self.refr is None:
            self.refr = np.tile(np.array([0, 1], np.float32), (self.geom4.shape[0], 1))
        if self.cube_box is None:
            self.cube_box = np.zeros((0, 6), np.float32)
            self.cube_mat4 = np.zeros((0, 4), np.float32)
            self.cube_spec = np.zeros((0,), np.float32)
            self.cube_refr = np.zeros((0, 2), np.float32)
        if self.order is None:
            self.order = np.arange(self.geom4.shape[0] + self.cube_box.shape[0], dtype=np.int32)

    @property
    def has_refraction(self) -> bool:
        return bool(np.any(self.refr[:, 0] > 0) or np.any(self.cube_refr[:, 0] > 0))

    @property
    def n_cubes(self) -> int:
        return int(self.cube_box.shape[0])

    def list_index(self, hit: np.ndarray) -> np.ndarray:
        """Device hit indices (spheres first, then cubes; -1 = miss) -> positions in the scene's object list."""
        hit = np.asarray(hit)
        table = np.concatenate([self.order.astype(hit.dtype), np.array([-1], hit.dtype)])
        return table[hit]            # -1 indexes the appended -1

    def objects_in_list_order(self) -> dict:
        """The object list as unified arrays in the reference's order (for the CPU oracle): kind 0 sphere
        (p0 = centre, radius), 1 cube (p0 = min_point, p1 = max_point)."""
        n, m = self.n_spheres, self.n_cubes
        inv = np.argsort(self.order, kind="stable")          # list position -> packed index
        kind = np.concatenate([np.zeros(n, np.int32), np.ones(m, np.int32)])[inv]
        p0 = np.concatenate([self.geom4[:, :3], self.cube_box[:, :3]])[inv]
        p1 = np.concatenate([np.zeros((n, 3), np.float32), self.cube_box[:, 3:]])[inv]
        radius = np.concatenate([self.radius, np.zeros(m, np.float32)])[inv]
        mat4 = np.concatenate([self.mat4, self.cube_mat4])[inv]
        spec = np.concatenate([self.spec, self.cube_spec])[inv]
        refr = np.concatenate([self.refr, self.cube_refr])[inv]
        return dict(kind=kind, p0=p0, p1=p1, radius=radius, mat4=mat4, spec=spec, refr=refr)

    @property
    def n_spheres(self) -> int:
        return int(self.geom4.shape[0])

    def ct_lights(self) -> RtLights:
        L = RtLights()
        v = [float(x) for x in self.lights]
        L.point_pos[:] = v[0:3]; L.point_color[:] = v[3:6]; L.point_intensity = v[6]
        L.area_pos[:] = v[7:10]; L.area_u[:] = v[10:13]; L.area_v[:] = v[13:16]
        L.area_color[:] = v[16:19]; L.area_intensity = v[19]
        L.area_samples = int(self.area_samples)
        return L


def _v3(v) -> tuple:
    return (float(v.x), float(v.y), float(v.z))


def pack_scene(scene: dict) -> PackedScene:
    """Flatten a reference-style scene dict into the SoA device layout.

    Values are rounded to float32 here, once; ``r*r`` is formed in float32 from the float32
    radius so the kernel sees exactly what :func:`scene_from_arrays` hands the FP64 oracle.
    Errors mirror what the reference would hit: a missing key raises ``KeyError``
    (raytracer.py:156,176).  Members of ``scene['spheres']`` with ``min_point`` / ``max_point`` are
    cubes (the reference's ``else`` branch, :159-160); anything else raises ``TypeError``.
    """
    objects = scene["spheres"]
    if len(objects) > MAX_SPHERES:
        raise ValueError(f"scene has {len(objects)} objects; this build supports at most {MAX_SPHERES}")
    is_cube = []
    for i, s in enumerate(objects):
        if hasattr(s, "center") and hasattr(s, "radius"):
            is_cube.append(False)
        elif hasattr(s, "min_point") and hasattr(s, "max_point"):
            is_cube.append(True)
        else:
            raise TypeError(f"scene['spheres'][{i}] is neither a sphere nor a cube")
    sph_pos = [i for i, c in enumerate(is_cube) if not c]
    cube_pos = [i for i, c in enumerate(is_cube) if c]
    n, nc = len(sph_pos), len(cube_pos)
    if n + 2 * nc > MAX_SPHERES:     # rt_set_cubes' limit (a cube takes two entries of the LDS object list): fail here,
        raise ValueError(            # before rt_upload_scene has replaced the scene on the device
            f"scene has {n} spheres and {nc} cubes; spheres + 2 x cubes must not exceed {MAX_SPHERES}")
    geom = np.zeros((n, 4), np.float32)
    mat = np.zeros((n, 4), np.float32)
    spec = np.zeros((n,), np.float32)
    rad = np.zeros((n,), np.float32)
    refr = np.zeros((n, 2), np.float32)
    cbox = np.zeros((nc, 6), np.float32)
    cmat = np.zeros((nc, 4), np.float32)
    cspec = np.zeros((nc,), np.float32)
    crefr = np.zeros((nc, 2), np.float32)

    def material(pos, m, mat_row, spec_arr, refr_arr, k):
        refr_arr[k] = (float(getattr(m, "refraction", 0)), float(getattr(m, "refractive_index", 1)))
        if refr_arr[k, 0] > 0 and not refr_arr[k, 1] > 0:
            raise ValueError(f"scene['spheres'][{pos}]: refractive_index must be > 0")
        mat_row[k, :3] = _v3(m.color)
        mat_row[k, 3] = m.reflection
        spec_arr[k] = m.specular

    for i, pos in enumerate(sph_pos):
        s = objects[pos]
        material(pos, s.material, mat, spec, refr, i)
        r32 = np.float32(s.radius)
        geom[i, :3] = _v3(s.center)
        geom[i, 3] = r32 * r32
        rad[i] = r32
    for j, pos in enumerate(cube_pos):
        c = objects[pos]
        material(pos, c.material, cmat, cspec, crefr, j)
        cbox[j, :3] = _v3(c.min_point)
        cbox[j, 3:] = _v3(c.max_point)
    pl, al = scene["global_light"], scene["area_light"]
    ns = int(al.samples)
    if not (1 <= ns <= MAX_AREA_SAMPLES):
        raise ValueError(f"area_light.samples must be in [1, {MAX_AREA_SAMPLES}], got {ns}")
    lights = np.array([*_v3(pl.position), *_v3(pl.color), float(pl.intensity),
                       *_v3(al.position), *_v3(al.u), *_v3(al.v), *_v3(al.color),
                       float(al.intensity)], dtype=np.float32)
    for name, a in (("geom4", geom), ("mat4", mat), ("spec", spec), ("lights", lights), ("refr", refr),
                    ("cube_box", cbox), ("cube_mat4", cmat), ("cube_spec", cspec), ("cube_refr", crefr)):
        if not np.all(np.isfinite(a)):
            raise ValueError(f"scene has non-finite values in {name}")
    return PackedScene(geom, mat, spec, rad, lights, ns, refr, cbox, cmat, cspec, crefr,
                       np.array(sph_pos + cube_pos, np.int32))


def scene_from_arrays(center, radius, color, specular, reflection, lights=None,
                      area_samples=12, refraction=None, refractive_index=None, kind=None, box_max=None) -> dict:
    """Build a reference-style scene dict from arrays (float32 values widened to Python floats),
    so oracle and kernel see bit-identical inputs.  ``kind[i] == 1`` makes object i a Cube with
    ``min_point = center[i]`` and ``max_point = box_max[i]`` (radius ignored)."""
    center = np.asarray(center, np.float32); radius = np.asarray(radius, np.float32)
    color = np.asarray(color, np.float32)
    specular = np.asarray(specular, np.float32); reflection = np.asarray(reflection, np.float32)
    n = len(radius)
    refraction = np.zeros(n, np.float32) if refraction is None else np.asarray(refraction, np.float32)
    refractive_index = np.ones(n, np.float32) if refractive_index is None else np.asarray(refractive_index, np.float32)
    kind = np.zeros(n, np.int32) if kind is None else np.asarray(kind, np.int32)
    box_max = None if box_max is None else np.asarray(box_max, np.float32)

    def obj(i):
        m = Material(Vector3(*map(float, color[i])), specular=float(specular[i]),
                     reflection=float(reflection[i]), refraction=float(refraction[i]),
                     refractive_index=float(refractive_index[i]))
        if kind[i] == 1:
            return Cube(Vector3(*map(float, center[i])), Vector3(*map(float, box_max[i])), m)
        return Sphere(Vector3(*map(float, center[i])), float(radius[i]), m)

    sph = [obj(i) for i in range(n)]
    d = default_scene()
    if lights is not None:
        v = [float(x) for x in np.asarray(lights, np.float32)]
        d["global_light"] = Light(Vector3(*v[0:3]), Vector3(*v[3:6]), v[6])
        d["area_light"] = AreaLight(Vector3(*v[7:10]), Vector3(*v[10:13]), Vector3(*v[13:16]),
                                    Vector3(*v[16:19]), v[19], samples=int(area_samples))
    d["spheres"] = sph
    return d


def generate_scene(n_spheres: int, seed: int, width: int = 1920, height: int = 1080,
                   keep_default: bool = False) -> dict:
    """Deterministic synthetic scenes for BASELINE.json configs 2-5.

    ``numpy.random.default_rng(seed)`` (PCG64); centre inside the view frustum
    (``z ~ U(-24,-4)``, ``x ~ U(-1,1)*|z|*0.9``, ``y ~ U(-1,1)*|z|*0.9*H/W``); radius
    ``U(0.3,1.2)*(8/N)^(1/3)`` clipped to >= 0.08; colour ``U(0.2,1)^3``; specular 1;
    reflection ``U(0.2,0.8)``; spheres overlapping an accepted sphere, the camera origin or a
    light are rejected.  ``keep_default`` starts from the reference's three spheres (config 2:
    default 3 + 5 generated).  Every value is rounded to float32.  Lights are the reference's.
    """
    rng = np.random.default_rng(seed)
    base = default_scene()
    centers, radii, colors, refl = [], [], [], []
    if keep_default:
        for s in base["spheres"]:
            centers.append(np.array(_v3(s.center))); radii.append(float(s.radius))
            colors.append(np.array(_v3(s.material.color))); refl.append(float(s.material.reflection))
    keep = [np.zeros(3), np.array(_v3(base["global_light"].position)),
            np.array(_v3(base["area_light"].position))]
    scale = (8.0 / max(n_spheres, 1)) ** (1.0 / 3.0)
    tries = 0
    while len(radii) < n_spheres:
        tries += 1
        if tries > 2_000_000:
            raise RuntimeError("scene generator could not place all spheres")
        z = rng.uniform(-24.0, -4.0)
        x = rng.uniform(-1.0, 1.0) * abs(z) * 0.9
        y = rng.uniform(-1.0, 1.0) * abs(z) * 0.9 * height / width
        r = max(0.08, rng.uniform(0.3, 1.2) * scale)
        col = rng.uniform(0.2, 1.0, 3)
        rf = rng.uniform(0.2, 0.8)
        c = np.array([x, y, z])
        if any(np.linalg.norm(c - k) <= r * 1.05 for k in keep):
            continue
        if centers:
            d = np.linalg.norm(np.array(centers) - c, axis=1)
            if np.any(d <= (np.array(radii) + r) * 1.02):
                continue
        centers.append(c); radii.append(r); colors.append(col); refl.append(rf)
    lights = pack_scene(base).lights
    return scene_from_arrays(np.array(centers), np.array(radii), np.array(colors),
                             np.ones(len(radii)), np.array(refl), lights, 12)


def orbit_camera(frame: int, n_frames: int = 256, radius: float = 6.0,
                 center=(0.0, 0.0, -6.0), height: float = 1.0) -> Camera:
    """Config 5: camera on a circle of ``radius`` around ``center`` at
    ``height``, looking at the centre; frame f at angle 2*pi*f/n_frames. Frame 0 sits at
    +z of the centre, i.e. near the reference's fixed camera position."""
    ang = 2.0 * math.pi * frame / n_frames
    c = np.array(center, np.float64)
    o = c + np.array([radius * math.sin(ang), height, radius * math.cos(ang)])
    f = c - o
    f /= np.linalg.norm(f)
    r = np.cross(f, np.array([0.0, 1.0, 0.0]))
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    return Camera(tuple(np.float32(o).tolist()), tuple(np.float32(r).tolist()),
                  tuple(np.float32(u).tolist()), tuple(np.float32(f).tolist()))


def refractive_scene() -> dict:
    """The reference's default scene with glass added (the reference itself only
    ships refraction in a commented-out cube, raytracer.py:88-89): the front sphere becomes partly
    refractive, and two glass spheres stand between the camera / the lights and the others, so that
    refracted paths, total internal reflection candidates and the "shadow ray's nearest hit is
    refractive" rule (raytracer.py:187, :202) are all exercised."""
    s = default_scene()
    m = s["spheres"][0].material
    m.refraction, m.refractive_index = 0.6, 1.5
    s["spheres"].append(Sphere(Vector3(-0.9, 0.6, -3.2), 0.45,
                               Material(Vector3(0.9, 0.95, 1.0), specular=0.8, reflection=0.1,
                                        refraction=0.85, refractive_index=1.33)))
    s["spheres"].append(Sphere(Vector3(1.6, 1.4, -3.5), 0.6,
                               Material(Vector3(1.0, 0.9, 0.8), specular=0.5, reflection=0.0,
                                        refraction=0.7, refractive_index=0.8)))
    return s


def cube_scene(glass: bool = False) -> dict:
    """The reference's default scene plus cubes in its object list (the reference
    only ships cubes in comments, raytracer.py:86-91), placed first, in the middle and last in the list
    so that the list order (:156) matters.

    The cubes are small on purpose.  intersect_cube returns ``t_max`` unchecked when ``t_min <= 0``
    (:137) and find_nearest_intersection accepts any non-zero ``t`` (:162), so a cube is "hit" by
    every ray whose LINE passes through it, in front of the origin or behind it: an opaque cube
    shades itself black (its own shadow rays leave a face and hit the cube at t < 0) and darkens every
    point whose line to a light passes through it in either direction.  A large cube would blacken
    the whole frame; these three leave structure to compare: one between the lights and the spheres
    (forward shadows), one in view (``glass``: the commented-out cube's material, :88-89, which the
    nearest-hit shadow rule :187 keeps lit), one behind the spheres as seen from the lights (backward
    shadows)."""
    s = default_scene()
    sph = s["spheres"]
    shown = Material(Vector3(1, 1, 1), specular=0.7, reflection=0.1, refraction=0.9, refractive_index=1.5) \
        if glass else Material(Vector3(0.9, 0.8, 0.3), specular=0.7, reflection=0.1)
    s["spheres"] = [
        Cube(Vector3(-1.5, -0.35, -4.3), Vector3(-0.9, 0.25, -3.7), shown),
        sph[0], sph[1],
        Cube(Vector3(1.1, 1.9, -2.9), Vector3(1.7, 2.5, -2.3),
             Material(Vector3(0.8, 0.8, 0.9), specular=0.5, reflection=0.5)),
        sph[2],
        Cube(Vector3(-4.2, -3.8, -13.0), Vector3(-2.6, -2.4, -11.4),
             Material(Vector3(0.8, 0.8, 0.8), specular=0.1, reflection=0.1)),
    ]
    return s
