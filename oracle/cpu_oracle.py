"""TEST INFRASTRUCTURE ONLY: ctypes front end of oracle/librt_oracle.so (oracle/rt_oracle.cpp)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librt_oracle.so")
_lib = None

DEFAULT_CAMERA = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, -1]], np.float32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librt_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.rto_render_pixels.restype = ctypes.c_int
        _lib.rto_render_pixels2.restype = ctypes.c_int
        _lib.rto_render_objects.restype = ctypes.c_int
        _lib.rto_max_threads.restype = ctypes.c_int
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def render_pixels(center, radius, color, specular, reflection, lights, area_samples, width, height,
                  pixels=None, samples=1, max_depth=5, mode="K", seed=0, camera=None,
                  precision="f64", threads=0, refraction=None, refractive_index=None, kind=None,
                  box_max=None):
    """Render the listed (x, y) pixels (or the full frame if ``pixels`` is None).

    ``kind`` / ``box_max``: objects in the reference's list order, kind 0 = sphere (center, radius),
    1 = cube (``center`` holds min_point, ``box_max`` max_point); None = all spheres.

    Scene arrays are float32 (the values the kernel gets); the oracle widens them to double.
    Returns dict(color f64[n,3], rgb8 u8[n,3], hit i32[n], trace_levels i64[n], hit_levels i64[n],
    margin f64[n]).
    """
    L = lib()
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    center, radius, color = f32(center).reshape(-1, 3), f32(radius).ravel(), f32(color).reshape(-1, 3)
    specular, reflection, lights = f32(specular).ravel(), f32(reflection).ravel(), f32(lights).ravel()
    n = len(radius)
    assert center.shape == (n, 3) and color.shape == (n, 3) and len(specular) == n == len(reflection)
    assert lights.shape == (20,)
    refraction = None if refraction is None else f32(refraction).ravel()
    refractive_index = None if refractive_index is None else f32(refractive_index).ravel()
    assert refraction is None or len(refraction) == n
    assert refractive_index is None or len(refractive_index) == n
    kind = None if kind is None else np.ascontiguousarray(kind, np.int32).ravel()
    box_max = None if box_max is None else f32(box_max).reshape(-1, 3)
    assert kind is None or (len(kind) == n and box_max is not None and box_max.shape == (n, 3))
    cam = f32(DEFAULT_CAMERA if camera is None else camera).reshape(12)
    if pixels is None:
        px, npx = None, width * height
    else:
        px = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
        npx = len(px)
    out = dict(color=np.zeros((npx, 3)), rgb8=np.zeros((npx, 3), np.uint8),
               hit=np.zeros(npx, np.int32), trace_levels=np.zeros(npx, np.int64),
               hit_levels=np.zeros(npx, np.int64), margin=np.zeros(npx))
    rc = L.rto_render_objects(
        ctypes.c_int(n), _p(kind, ctypes.c_int32), _p(box_max, ctypes.c_float), _p(center, ctypes.c_float), _p(radius, ctypes.c_float),
        _p(color, ctypes.c_float), _p(specular, ctypes.c_float), _p(reflection, ctypes.c_float),
        _p(refraction, ctypes.c_float), _p(refractive_index, ctypes.c_float),
        _p(lights, ctypes.c_float), ctypes.c_int(int(area_samples)), _p(cam, ctypes.c_float),
        ctypes.c_int(width), ctypes.c_int(height), _p(px, ctypes.c_int32), ctypes.c_int64(npx),
        ctypes.c_int(samples), ctypes.c_int(max_depth), ctypes.c_int({"C": 0, "K": 1}[mode]),
        ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_int({"f64": 0, "f32": 1}[precision]),
        ctypes.c_int(threads), _p(out["color"], ctypes.c_double), _p(out["rgb8"], ctypes.c_uint8),
        _p(out["hit"], ctypes.c_int32), _p(out["trace_levels"], ctypes.c_int64),
        _p(out["hit_levels"], ctypes.c_int64), _p(out["margin"], ctypes.c_double))
    if rc != 0:
        raise ValueError(f"rto_render_pixels failed with code {rc}")
    return out


def render_scene(scene: dict, width, height, **kw):
    """Same, from a raytracer_amd-style scene dict (uses raytracer_amd.scenes only to flatten it)."""
    from raytracer_amd.scenes import PackedScene, pack_scene
    p = scene if isinstance(scene, PackedScene) else pack_scene(scene)
    cam = kw.pop("camera", None)
    if cam is not None and hasattr(cam, "as_array"):
        cam = cam.as_array()
    o = p.objects_in_list_order()      # hit indices are positions in the scene's object list
    return render_pixels(o["p0"], o["radius"], o["mat4"][:, :3], o["spec"], o["mat4"][:, 3], p.lights,
                         p.area_samples, width, height, camera=cam, refraction=o["refr"][:, 0],
                         refractive_index=o["refr"][:, 1], kind=o["kind"], box_max=o["p1"], **kw)


def max_threads() -> int:
    return int(lib().rto_max_threads())
