"""raytracer_amd — MI355X-native implementation of aram-ap/RayTracer's per-pixel render path.

Public surface = the reference's: ``render``, ``render_chunk``, the scene classes and the
module-global ``scene`` (raytracer.py:10-87, 299-325).  The compute lives in
``csrc/rt355.hip`` behind the C ABI in ``include/rt355.h``.
"""
from .scenes import (AreaLight, Camera, Light, Material, Sphere, Vector3, default_scene,  # noqa: F401
                    generate_scene, orbit_camera, pack_scene, scene_from_arrays)
from . import render as _render_mod
from .render import Renderer, render, render_chunk  # noqa: F401


def __getattr__(name):
    if name == "scene":            # the reference's module-global scene (raytracer.py:78)
        return _render_mod.scene
    raise AttributeError(name)


def set_scene(scene: dict) -> None:
    """Replace the module-global scene that ``render`` reads (raytracer.py:317 reads a global)."""
    _render_mod.scene = scene
