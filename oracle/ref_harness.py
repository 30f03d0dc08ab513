"""TEST INFRASTRUCTURE ONLY — never imported by the product path (raytracer_amd/).

Imports the *reference itself* (``/root/reference/raytracer.py``) in this container and runs it
under a position-keyed RNG so that its per-pixel results can be pinned as golden vectors
(SURVEY.md §2.2, §8c).  The reference source is never copied: it is loaded from where it lies.
``/root/reference`` does not exist on the GPU box, so nothing under ``tests/ -m gpu``, ``smoke()``
or ``bench.py`` imports this module; only ``oracle/gen_golden.py`` and the CPU-side oracle
pinning tests (skipped when the reference is absent) do.

What is patched, and only via module attributes (the reference resolves its globals at call
time, raytracer.py:180,185,248,257,296):

* ``OpenGL*`` — empty stub modules (PyOpenGL is not installed; raytracer.py:5-7 star-imports
  them, the names are only used by display()/reshape()/main(), which are never called here).
* ``random`` — replaced by a shim whose ``random()`` is a pure function of
  (seed, x, y, sample, level, draw index)  [mode K]  or the constant 0.5  [mode C].
* ``trace_ray`` — wrapped to (a) start the recursion at ``depth = 5 - D`` so that the literal
  ``depth > 5`` cut-off (raytracer.py:233) acts as ``level > D``, (b) track the logical level
  for the RNG key, (c) record the primary hit index and count trace levels.
* ``compute_lighting`` — wrapped to reset the per-call draw counter and count shaded hits.

Keyed RNG definition (restated identically in oracle/rt_oracle.cpp and the HIP kernel):

    fmix32(h): h ^= h>>16; h *= 0x85EBCA6B; h ^= h>>13; h *= 0xC2B2AE35; h ^= h>>16      (uint32 arithmetic)
    pixel_key = fmix32(seed32 + x*0x9E3779B9 + y*0x85EBCA6B + sample*0xC2B2AE35)
    level_key = fmix32(pixel_key + 0x9E3779B9 + L)      L = 1 + logical depth, for compute_lighting's draws;
                                                        the two jitter draws of render_pixel use pixel_key itself
    pair(key, j) = g(key + j*0x9E3779B9 + 0x7F4A7C15)   g(h): h *= 0x85EBCA6B; h ^= h>>13; h *= 0xC2B2AE35; h ^= h>>16
    one hash gives TWO draws of 16 bits: (pair >> 16) * 2**-16 and (pair & 0xFFFF) * 2**-16, both in [0, 1)
    jitter: pair(pixel_key, 0) -> (u draw, v draw);  area-light sample s: pair(level_key, s) -> (random_u, random_v)
(Round 2 definition: a third of the arithmetic of round 1's one-fmix32-per-24-bit-draw stream, which was a
quarter of the shading kernel's instructions; every fixture was rendered again by the reference under it.)
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys
import types

import numpy as np

REFERENCE_PATH = os.environ.get("RT_REFERENCE_PATH", "/root/reference/raytracer.py")
M32 = 0xFFFFFFFF


def reference_available() -> bool:
    return os.path.isfile(REFERENCE_PATH)


def fmix32(h: int) -> int:
    h &= M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def mix(h: int, v: int) -> int:
    return fmix32((h + 0x9E3779B9 + v) & M32)


def pixel_key(seed32: int, x: int, y: int, sample: int) -> int:
    return fmix32((seed32 + x * 0x9E3779B9 + y * 0x85EBCA6B + sample * 0xC2B2AE35) & M32)


def pair_hash(key: int, j: int) -> int:
    h = (key + j * 0x9E3779B9 + 0x7F4A7C15) & M32
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def draw_value(key: int, i: int) -> float:
    """Draw i of the stream `key`: the first (i even) or second (i odd) half of pair i // 2."""
    h = pair_hash(key, i >> 1)
    return ((h >> 16) if (i & 1) == 0 else (h & 0xFFFF)) * 2.0 ** -16


def fold_seed(seed: int) -> int:
    seed &= 0xFFFFFFFFFFFFFFFF
    return (seed ^ (seed >> 32)) & M32


class _Ctx:
    """Mutable state shared by the shim and the wrappers."""

    def __init__(self):
        self.mode = "C"
        self.seed32 = 0
        self.D = 5
        self.x = self.y = 0
        self.jitter_draws = 0
        self.sample = 0
        self.level = 0            # logical depth of the trace_ray call in progress
        self.in_lighting = False
        self.draw_idx = 0
        self.primary_hit = -1
        self.trace_levels = 0     # trace_ray calls that scanned the scene (level <= D)
        self.hit_levels = 0       # compute_lighting calls


class _RandomShim:
    def __init__(self, ctx: _Ctx):
        self.ctx = ctx

    def random(self) -> float:
        c = self.ctx
        if c.in_lighting:
            i = c.draw_idx
            c.draw_idx += 1
            level = 1 + c.level
        else:
            c.sample = c.jitter_draws // 2
            i = c.jitter_draws % 2
            c.jitter_draws += 1
            level = 0
        if c.mode == "C":
            return 0.5
        pk = pixel_key(c.seed32, c.x, c.y, c.sample)
        return draw_value(pk if level == 0 else mix(pk, level), i)


_REF = None
_CTX = _Ctx()


def load_reference():
    """Import the reference module once, with stubs and wrappers installed."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise FileNotFoundError(REFERENCE_PATH)
    for name in ("OpenGL", "OpenGL.GL", "OpenGL.GLUT", "OpenGL.GLU"):
        sys.modules.setdefault(name, types.ModuleType(name))
    root = logging.getLogger()
    saved = (root.level, list(root.handlers))
    spec = importlib.util.spec_from_file_location("_rt_reference", REFERENCE_PATH)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    # undo the import-time logging.basicConfig(DEBUG) (raytracer.py:287) and silence tqdm (:315)
    root.handlers[:] = saved[1]
    root.setLevel(saved[0])
    ref.logging = types.SimpleNamespace(info=lambda *a, **k: None, debug=lambda *a, **k: None)
    ref.tqdm = lambda it, **k: it

    ctx = _CTX
    orig_trace, orig_light = ref.trace_ray, ref.compute_lighting
    orig_find = ref.find_nearest_intersection

    def trace_ray(ray, scene, depth=None):
        if depth is None:                       # top-level call from render_pixel (:296)
            depth = 5 - ctx.D
        level = depth - (5 - ctx.D)
        if level == 0:
            hit = orig_find(ray, scene)
            ctx.primary_hit = -1 if hit is None else next(
                i for i, o in enumerate(scene["spheres"]) if o is hit[0])
        if depth <= 5:
            ctx.trace_levels += 1
        prev = ctx.level
        ctx.level = level
        try:
            return orig_trace(ray, scene, depth)
        finally:
            ctx.level = prev

    def compute_lighting(point, normal, view_dir, material, scene):
        ctx.in_lighting = True
        ctx.draw_idx = 0
        ctx.hit_levels += 1
        try:
            return orig_light(point, normal, view_dir, material, scene)
        finally:
            ctx.in_lighting = False

    ref.trace_ray = trace_ray
    ref.compute_lighting = compute_lighting
    ref.random = _RandomShim(ctx)
    ref._orig = types.SimpleNamespace(trace_ray=orig_trace, compute_lighting=orig_light,
                                      find_nearest_intersection=orig_find)
    _REF = ref
    return ref


def to_ref_scene(ref, scene: dict) -> dict:
    """Rebuild a raytracer_amd-style scene dict out of the reference's own classes, with every
    value rounded to float32 (what pack_scene hands the kernel and the C++ oracle).  Pass
    ``scene=None`` to render_pixels to use the reference's own unrounded module-global scene."""
    V = ref.Vector3

    def f(a):        # every scene value is rounded to float32 first: the kernel's inputs
        return float(np.float32(a))

    def v(a):
        return V(f(a.x), f(a.y), f(a.z))

    def mat(m):
        return ref.Material(v(m.color), specular=f(m.specular), reflection=f(m.reflection),
                            refraction=f(getattr(m, "refraction", 0)),
                            refractive_index=f(getattr(m, "refractive_index", 1)))

    def obj(s):      # the reference's own classes: find_nearest_intersection dispatches on them (:157)
        if hasattr(s, "min_point"):
            return ref.Cube(v(s.min_point), v(s.max_point), mat(s.material))
        return ref.Sphere(v(s.center), f(s.radius), mat(s.material))

    pl, al = scene["global_light"], scene["area_light"]
    return {
        "global_light": ref.Light(v(pl.position), v(pl.color), f(pl.intensity)),
        "area_light": ref.AreaLight(v(al.position), v(al.u), v(al.v), v(al.color),
                                    f(al.intensity), samples=int(al.samples)),
        "spheres": [obj(s) for s in scene["spheres"]],
    }


def render_pixels(scene: dict | None, width: int, height: int, pixels, samples: int = 1,
                  max_depth: int = 5, mode: str = "K", seed: int = 0, camera=None):
    """Run the reference's render_pixel (raytracer.py:289-297) on the listed (x, y) pixels.

    Returns dict of arrays: ``color`` float64 [n,3] unclamped, ``rgb8`` uint8 [n,3] quantised as
    render() does (:318,:325), ``hit`` int32 [n] primary hit index of the LAST sample (-1 = miss),
    ``trace_levels``/``hit_levels`` int64 [n] (rays = trace_levels + 13*hit_levels for 12 area
    samples).

    ``scene=None`` uses the reference's own module-global scene (:78-87).  ``camera`` (a
    raytracer_amd.scenes.Camera) is a harness-side extension: the reference has no camera, so for
    a non-default camera the primary ray is built here with the reference's Vector3/Ray classes
    following :293-295 and handed to the reference's trace_ray.
    """
    ref = load_reference()
    ctx = _CTX
    ctx.mode, ctx.seed32, ctx.D = mode, fold_seed(seed), int(max_depth)
    rscene = ref.scene if scene is None else to_ref_scene(ref, scene)
    pixels = np.asarray(pixels, np.int64).reshape(-1, 2)
    n = len(pixels)
    out = dict(color=np.zeros((n, 3)), rgb8=np.zeros((n, 3), np.uint8),
               hit=np.full(n, -1, np.int32), trace_levels=np.zeros(n, np.int64),
               hit_levels=np.zeros(n, np.int64))
    cam = None
    if camera is not None:
        a = camera.as_array().astype(np.float64)
        is_default = np.array_equal(a, np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, -1.0]]))
        cam = None if is_default else [ref.Vector3(*map(float, r)) for r in a]
    for k, (x, y) in enumerate(pixels):
        x, y = int(x), int(y)
        ctx.x, ctx.y, ctx.jitter_draws, ctx.sample = x, y, 0, 0
        ctx.trace_levels = ctx.hit_levels = 0
        ctx.primary_hit = -1
        if cam is None:
            col = ref.render_pixel(x, y, width, height, rscene, samples)
        else:
            col = _render_pixel_camera(ref, cam, x, y, width, height, rscene, samples)
        c = (col.x, col.y, col.z)
        out["color"][k] = c
        clamped = np.array([min(1, max(0, ch)) for ch in c])        # :318
        out["rgb8"][k] = (clamped * 255).astype(np.uint8)           # :325
        out["hit"][k] = ctx.primary_hit
        out["trace_levels"][k] = ctx.trace_levels
        out["hit_levels"][k] = ctx.hit_levels
    return out


def _render_pixel_camera(ref, cam, x, y, width, height, scene, samples):
    """render_pixel (:289-297) with the fixed pinhole replaced by a camera frame."""
    o, r, u_, f = cam
    aspect_ratio = width / height
    color = ref.Vector3(0, 0, 0)
    for _ in range(samples):
        u = ((x + ref.random.random()) / width) * 2 - 1
        v = -((y + ref.random.random()) / height * 2 - 1) / aspect_ratio
        d = (r * u + u_ * v + f).normalize()
        color = color + ref.trace_ray(ref.Ray(o, d), scene)
    return color * (1 / samples)


def render_full(scene, width, height, samples=1, max_depth=5, mode="K", seed=0, camera=None):
    """Full frame through render_pixels; returns image uint8 [H,W,3] plus per-pixel maps."""
    ys, xs = np.mgrid[0:height, 0:width]
    px = np.stack([xs.ravel(), ys.ravel()], 1)
    r = render_pixels(scene, width, height, px, samples, max_depth, mode, seed, camera)
    return {k: v.reshape(height, width, *v.shape[1:]) for k, v in r.items()}


def reference_render_native(width, height, samples=1, seed=0):
    """The reference's own render() (:310-325), untouched except for a seeded Mersenne Twister
    (random.seed) — used to check this harness against SURVEY.md §8c's recorded SHA-1."""
    import random as pyrandom
    ref = load_reference()
    saved = (ref.random, ref.trace_ray, ref.compute_lighting)
    ref.random, ref.trace_ray, ref.compute_lighting = (
        pyrandom, ref._orig.trace_ray, ref._orig.compute_lighting)
    try:
        pyrandom.seed(seed)
        return ref.render(width, height, samples)
    finally:
        ref.random, ref.trace_ray, ref.compute_lighting = saved
