"""ctypes binding of librt355.so (include/rt355.h).  Fails loudly when the library is missing:
there is no CPU path behind this package."""
from __future__ import annotations

import ctypes
import os

from .scenes import RtCamera, RtLights

_SO = os.environ.get("RT355_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                   "librt355.so")   # RT355_LIB: A/B builds only
_lib = None

RT_RNG_CONSTANT, RT_RNG_KEYED = 0, 1
FRAMES_PER_LAUNCH = 1024   # RT_FRAMES_PER_LAUNCH of include/rt355.h
TILE_ENTRY_BYTES = 196     # RT_TILE_ENTRY_BYTES
COMM_ID_BYTES = 128        # RT_COMM_ID_BYTES
RT_OPT_KERNEL, RT_KERNEL_STREAM, RT_KERNEL_PIXEL = 1, 0, 1
RT_OPT_OCCLUDER_TABLE = 2
RT_OPT_QUEUE_SPILL = 4
RT_OPT_TILE_FLUSH = 5
RT_OPT_REFLECT_TABLE = 6
RNG_MODES = {"C": RT_RNG_CONSTANT, "constant": RT_RNG_CONSTANT, RT_RNG_CONSTANT: RT_RNG_CONSTANT,
             "K": RT_RNG_KEYED, "keyed": RT_RNG_KEYED}

#: every symbol include/rt355.h declares (tests check the .so exports all of them)
SYMBOLS = ("rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_synchronize",
           "rt_set_option", "rt_upload_scene", "rt_set_refraction", "rt_set_cubes", "rt_occluder_table", "rt_occluder_table_buckets", "rt_reflect_table", "rt_render", "rt_render_noalloc", "rt_render_rows", "rt_render_frames_rows",
           "rt_render_chunk", "rt_render_frames",
           "rt_scatter_rows", "rt_scatter_rows_frames", "rt_compact_tiles", "rt_place_tiles",
           "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_comm_info", "rt_comm_all_gather_counts", "rt_comm_gather",
           "rt_get_ray_counts", "rt_reset_ray_counts", "rt_debug_counters", "rt_version")


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt355 error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return _SO


def load():
    """Load librt355.so and declare prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_SO):
        raise ImportError(f"{_SO} is missing: build it with `python -m raytracer_amd.build` "
                          "(hipcc --offload-arch=gfx950). rt355 has no CPU fallback.")
    # torch first: its wheel bundles its own libamdhip64; loading ours afterwards binds to that
    # already-loaded runtime (same SONAME) so the process has exactly one HIP runtime and
    # torch tensors / streams are valid inside rt355.
    import torch  # noqa: F401
    L = ctypes.CDLL(_SO)
    c_int, c_void_p, c_u64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64
    P = ctypes.POINTER
    L.rt_create.argtypes = [c_int, P(c_void_p)]
    L.rt_destroy.argtypes = [c_void_p]; L.rt_destroy.restype = None
    L.rt_last_error.argtypes = [c_void_p]; L.rt_last_error.restype = ctypes.c_char_p
    L.rt_set_stream.argtypes = [c_void_p, c_void_p, c_int]
    L.rt_synchronize.argtypes = [c_void_p]
    L.rt_upload_scene.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, P(RtLights)]
    L.rt_set_refraction.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.rt_occluder_table.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
    L.rt_occluder_table_buckets.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p]
    L.rt_reflect_table.argtypes = [c_void_p, c_int, c_int, c_void_p]
    L.rt_set_cubes.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]
    L.rt_render.argtypes = [c_void_p, P(RtCamera), c_int, c_int, c_int, c_int, c_int, c_u64,
                            c_void_p, c_void_p]
    L.rt_render_noalloc.argtypes = L.rt_render.argtypes
    L.rt_render_rows.argtypes = [c_void_p, P(RtCamera), c_int, c_int, c_int, c_int, c_int, c_u64,
                                 c_int, c_int, c_int, c_void_p, c_void_p, P(c_int)]
    L.rt_render_frames_rows.argtypes = [c_void_p, P(RtCamera), P(c_u64), c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p,
                                        P(c_int)]
    L.rt_set_option.argtypes = [c_void_p, c_int, c_int]
    L.rt_render_chunk.argtypes = [c_void_p, P(RtCamera), c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_u64, c_void_p]
    L.rt_render_frames.argtypes = [c_void_p, P(RtCamera), P(c_u64), c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p]
    L.rt_scatter_rows.argtypes = [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int,
                                  c_void_p]
    L.rt_scatter_rows_frames.argtypes = [c_void_p, c_void_p, ctypes.c_int64, ctypes.c_int64, c_int, c_int, c_int,
                                         c_int, c_int, c_void_p]
    L.rt_compact_tiles.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int]
    L.rt_place_tiles.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                 ctypes.c_int64, c_void_p]
    L.rt_comm_unique_id.argtypes = [c_void_p]
    L.rt_comm_init.argtypes = [c_void_p, c_void_p, c_int, c_int]
    L.rt_comm_destroy.argtypes = [c_void_p]
    L.rt_comm_info.argtypes = [c_void_p, P(c_int), P(c_int), P(c_int)]
    L.rt_comm_all_gather_counts.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    L.rt_comm_gather.argtypes = [c_void_p, c_void_p, ctypes.c_int64, c_void_p, P(ctypes.c_int64), ctypes.c_int64, c_int,
                                 c_void_p]
    L.rt_get_ray_counts.argtypes = [c_void_p, P(c_u64)]
    L.rt_reset_ray_counts.argtypes = [c_void_p]
    L.rt_debug_counters.argtypes = [c_void_p, P(c_u64)]
    L.rt_version.argtypes = []; L.rt_version.restype = ctypes.c_char_p
    for name in SYMBOLS:
        if name not in ("rt_destroy", "rt_last_error", "rt_version"):
            getattr(L, name).restype = c_int
    _lib = L
    return L


def check(rc: int, ctx=None):
    if rc != 0:
        msg = load().rt_last_error(ctx)
        raise RtError(rc, msg.decode() if msg else "unknown error")
