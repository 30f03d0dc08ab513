"""Host side of the rt355 render path: the reference's entry points over the HIP library.

Mirrors ``render(width, height, samples=1)`` (raytracer.py:310-325) and
``render_chunk(args)`` (raytracer.py:299-308) — same positional arguments, same return
type/shape/dtype/row order — and reads the module-global :data:`scene` exactly as the reference
does (raytracer.py:317) unless ``scene=`` is given.  Everything per-pixel happens in
``csrc/rt355.hip``; PyTorch is used only to own device buffers and streams.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .scenes import Camera, PackedScene, RtCamera, default_scene, pack_scene

#: module-global scene, as in the reference (raytracer.py:78-87); ``render`` reads it at call time
scene = default_scene()

_DEFAULT_MAX_DEPTH = 5      # the literal in trace_ray (raytracer.py:233)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("rt355 needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU path")
    return torch


class Renderer:
    """One GPU context (``rt_ctx``): scene upload once, many renders (animation frames reuse it)."""

    def __init__(self, device: int | None = None):
        torch = _torch()
        self._L = _lib.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = ctypes.c_void_p()
        _lib.check(self._L.rt_create(self.device, ctypes.byref(h)))
        self._h = h
        self._packed: PackedScene | None = None
        self._options = {"occluder_table": True, "reflect_table": True}     # options that take effect at upload time
        self._uploaded_options = None
        self.use_torch_stream()

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        _lib.check(rc, self._h)

    # -- stream / scene -------------------------------------------------------------------
    def use_torch_stream(self):
        """Launch on torch's current stream for this device, so torch ops, torch.cuda.Event timing
        and RCCL collectives order against the render kernels without extra syncs."""
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._L.rt_set_stream(self._h, ctypes.c_void_p(s), 0))
        self._stream = s

    def synchronize(self):
        self._check(self._L.rt_synchronize(self._h))

    def upload_scene(self, scn: dict | PackedScene | None = None, *, force: bool = False) -> PackedScene:
        p = scn if isinstance(scn, PackedScene) else pack_scene(scene if scn is None else scn)
        if not force and self._packed is not None and self._uploaded_options == self._options \
                and _same_scene(p, self._packed):
            return self._packed          # the device already holds exactly this scene (and its occluder table)
        lights = p.ct_lights()
        g, m, s = (np.ascontiguousarray(a, np.float32) for a in (p.geom4, p.mat4, p.spec))
        # From here on the device scene is in flux (rt_upload_scene resets it, rt_set_refraction / rt_set_cubes
        # complete it): the cache names no scene until all three have succeeded, so a failure midway can never make
        # a later render of the PREVIOUS scene skip its upload and trace the half-uploaded one.
        self._packed = None
        self._uploaded_options = None
        self._check(self._L.rt_upload_scene(self._h, g.ctypes.data, m.ctypes.data, s.ctypes.data,
                                            p.n_spheres, ctypes.byref(lights)))
        if p.has_refraction:      # Material.refraction / .refractive_index (raytracer.py:43, :213-230)
            f = np.ascontiguousarray(p.refr[:, 0], np.float32)
            ri = np.ascontiguousarray(p.refr[:, 1], np.float32)
            self._check(self._L.rt_set_refraction(self._h, f.ctypes.data, ri.ctypes.data, p.n_spheres))
        if p.n_cubes:             # Cube members of the object list (raytracer.py:56-60, :157-160)
            a = [np.ascontiguousarray(x, np.float32) for x in
                 (p.cube_box, p.cube_mat4, p.cube_spec, p.cube_refr[:, 0], p.cube_refr[:, 1])]
            self._check(self._L.rt_set_cubes(self._h, *[x.ctypes.data for x in a], p.n_cubes))
        self._packed = p
        self._uploaded_options = dict(self._options)
        return p

    # -- renders --------------------------------------------------------------------------
    def _alloc(self, shape, dtype):
        torch = _torch()
        return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")

    def render(self, width, height, samples=1, *, max_depth=_DEFAULT_MAX_DEPTH, camera=None,
               rng_mode="keyed", seed=0, out=None, return_hit_index=False):
        """Full frame -> uint8 tensor [H, W, 3] on the GPU (and int32 [H, W] hit index)."""
        torch = _torch()
        width, height = _dims(width, height)
        if out is None:
            out = self._alloc((height, width, 3), torch.uint8)
        _check_out(out, (height, width, 3), torch.uint8, self.device)
        hit = self._alloc((height, width), torch.int32) if return_hit_index else None
        cam = RtCamera.from_camera(camera)
        if width * height:
            self._check(self._L.rt_render(self._h, ctypes.byref(cam), width, height, int(samples),
                                          int(max_depth), _lib.RNG_MODES[rng_mode], int(seed),
                                          out.data_ptr(), hit.data_ptr() if hit is not None else None))
        p = self._packed
        if hit is not None and p is not None and not np.array_equal(p.order, np.arange(len(p.order))):
            # device index (spheres, then cubes) -> position in the scene's object list
            table = torch.tensor(list(p.order) + [-1], dtype=torch.int32, device=hit.device)
            hit = table[hit.long()]
        return (out, hit) if return_hit_index else out

    def render_rows(self, width, height, samples=1, *, tile_rows, tile_first, tile_stride,
                    max_depth=_DEFAULT_MAX_DEPTH, camera=None, rng_mode="keyed", seed=0, out=None):
        """Row tiles tile_first, tile_first+tile_stride, ... packed -> uint8 [rows, W, 3]."""
        torch = _torch()
        from .tiling import local_row_count
        width, height = _dims(width, height)
        rows = local_row_count(height, tile_rows, tile_first, tile_stride)
        if out is None:
            out = self._alloc((rows, width, 3), torch.uint8)
        if out.shape[0] < rows:
            raise ValueError(f"out holds {out.shape[0]} rows, {rows} needed")
        _check_out(out, (out.shape[0], width, 3), torch.uint8, self.device)
        cam = RtCamera.from_camera(camera)
        n = ctypes.c_int(0)
        self._check(self._L.rt_render_rows(self._h, ctypes.byref(cam), width, height, int(samples),
                                           int(max_depth), _lib.RNG_MODES[rng_mode], int(seed),
                                           int(tile_rows), int(tile_first), int(tile_stride),
                                           out.data_ptr(), None, ctypes.byref(n)))
        assert n.value == rows, (n.value, rows)
        return out[:rows]

    def render_chunk(self, x_start, x_end, y_start, y_end, width, height, samples=1, *,
                     max_depth=_DEFAULT_MAX_DEPTH, camera=None, rng_mode="keyed", seed=0):
        """Clamped float32 colour of a sub-rectangle -> tensor [y_end-y_start, x_end-x_start, 3]."""
        torch = _torch()
        h, w = int(y_end) - int(y_start), int(x_end) - int(x_start)
        if h < 0 or w < 0:
            raise ValueError("negative dimensions are not allowed")   # numpy's message (np.zeros)
        out = self._alloc((h, w, 3), torch.float32)
        cam = RtCamera.from_camera(camera)
        if h * w:
            self._check(self._L.rt_render_chunk(self._h, ctypes.byref(cam), int(x_start), int(x_end),
                                                int(y_start), int(y_end), int(width), int(height),
                                                int(samples), int(max_depth),
                                                _lib.RNG_MODES[rng_mode], int(seed), out.data_ptr()))
        return out

    @staticmethod
    def pack_cameras(cameras):
        """Cameras -> the rt_camera array render_frames hands to the library.  Animation loops that reuse
        their cameras can pack them once and pass the result instead of the list."""
        packed = {}
        items = []
        for c in cameras:
            k = id(c)
            if k not in packed:
                packed[k] = RtCamera.from_camera(c)
            items.append(packed[k])
        return (RtCamera * len(items))(*items)

    def render_frames(self, cameras, seeds, width, height, samples=1, *,
                      max_depth=_DEFAULT_MAX_DEPTH, rng_mode="keyed", out=None,
                      tile_rows=None, tile_first=0, tile_stride=1):
        """Animation batch -> uint8 [F, H, W, 3]; up to 1024 frames (RT_FRAMES_PER_LAUNCH) share one kernel launch pair.

        With ``tile_rows`` the batch holds only this rank's row tiles of every frame
        (-> uint8 [F, local_rows, W, 3], rt_render_frames_rows)."""
        torch = _torch()
        from .tiling import local_row_count
        width, height = _dims(width, height)
        n = len(cameras)
        if len(seeds) != n:
            raise ValueError("one seed per camera")
        if tile_rows is None:
            tile_rows, tile_first, tile_stride = max(height, 1), 0, 1
        rows = local_row_count(height, tile_rows, tile_first, tile_stride) if height else 0
        if out is None:
            out = self._alloc((n, rows, width, 3), torch.uint8)
        # `out` may hold more rows per frame than this rank owns (equal slabs for the gather)
        if out.dim() != 4 or out.shape[0] != n or out.shape[1] < rows:
            raise ValueError(f"out must be [n_frames, >= {rows}, {width}, 3]")
        _check_out(out, (n, out.shape[1], width, 3), torch.uint8, self.device)
        stride_px = out.shape[1] * width
        cams = cameras if isinstance(cameras, ctypes.Array) else self.pack_cameras(cameras)
        sd = (ctypes.c_uint64 * max(n, 1))(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds])
        if n and width * rows:
            nrows = ctypes.c_int(0)
            self._check(self._L.rt_render_frames_rows(
                self._h, cams, sd, n, width, height, int(samples), int(max_depth),
                _lib.RNG_MODES[rng_mode], int(tile_rows), int(tile_first), int(tile_stride),
                stride_px, out.data_ptr(), None, ctypes.byref(nrows)))
            assert nrows.value == rows
        return out[:, :rows]

    def set_kernel(self, name: str):
        """'stream' (default: LDS work-queue kernel) or 'pixel' (one thread per pixel baseline)."""
        v = {"stream": _lib.RT_KERNEL_STREAM, "pixel": _lib.RT_KERNEL_PIXEL}[name]
        self._check(self._L.rt_set_option(self._h, _lib.RT_OPT_KERNEL, v))

    def set_occluder_table(self, on: bool):
        """Shadow scan over the per-object candidate table (default) or over every sphere; same images.
        Takes effect at the next upload_scene."""
        self._check(self._L.rt_set_option(self._h, _lib.RT_OPT_OCCLUDER_TABLE, 1 if on else 0))
        self._options["occluder_table"] = bool(on)

    def set_reflect_table(self, on: bool):
        """Reflection rays of scenes with 4 or more spheres test only the spheres a per-object direction table lists
        (default; bit masks up to 255 spheres, lists from 256 on) or every sphere; same images.  Takes effect at the
        next upload_scene."""
        self._check(self._L.rt_set_option(self._h, _lib.RT_OPT_REFLECT_TABLE, 1 if on else 0))
        self._options["reflect_table"] = bool(on)

    def set_queue_spill(self, on):
        """Refractive scenes: a full work queue sets records aside in device memory (True, default) or only
        shrinks its passes (False); an int n >= 2 caps the stack at n blocks of 64 records per wave.  Same
        images up to float summation order."""
        self._check(self._L.rt_set_option(self._h, _lib.RT_OPT_QUEUE_SPILL, int(on)))

    def set_tile_flush(self, on: bool):
        """8-bit frames at one sample per pixel: the stream kernel stores whole tiles from LDS staging (True,
        default) or every finished pixel by itself (False); same bytes."""
        self._check(self._L.rt_set_option(self._h, _lib.RT_OPT_TILE_FLUSH, 1 if on else 0))

    def scatter_rows(self, gathered, n_ranks, width, height, tile_rows, out=None):
        """Root side of the multi-GPU gather: [n_ranks, stride] packed slabs -> image [H, W, 3]."""
        torch = _torch()
        if out is None:
            out = self._alloc((height, width, 3), torch.uint8)
        g = gathered
        if g.dim() != 2 or g.shape[0] != n_ranks or g.dtype != torch.uint8 or g.stride(1) != 1:
            raise ValueError("gathered must be a uint8 [n_ranks, slab_bytes] tensor with unit inner stride")
        _check_out(out, (height, width, 3), torch.uint8, self.device)
        self._check(self._L.rt_scatter_rows(self._h, g.data_ptr(), g.stride(0) if n_ranks > 1 else g.shape[1],
                                            n_ranks, width, height, tile_rows, out.data_ptr()))
        return out

    def scatter_rows_frames(self, gathered, n_ranks, width, height, tile_rows, out=None):
        """The same for a batch in one launch: [n_ranks, F, slab_bytes] -> [F, H, W, 3]."""
        torch = _torch()
        g = gathered
        if g.dim() != 3 or g.shape[0] != n_ranks or g.dtype != torch.uint8 or g.stride(2) != 1:
            raise ValueError("gathered must be a uint8 [n_ranks, n_frames, slab_bytes] tensor with unit inner stride")
        nf = g.shape[1]
        if out is None:
            out = self._alloc((nf, height, width, 3), torch.uint8)
        _check_out(out, (nf, height, width, 3), torch.uint8, self.device)
        if nf:
            self._check(self._L.rt_scatter_rows_frames(self._h, g.data_ptr(), g.stride(0), g.stride(1), n_ranks, nf,
                                                       width, height, tile_rows, out.data_ptr()))
        return out

    def compact_tiles(self, slab, entries, count):
        """Sender side of the sparse exchange (rt_compact_tiles): the occupied 8x8 tiles of the batch the last
        render call wrote into ``slab`` -> ``entries`` uint8 [capacity, TILE_ENTRY_BYTES], their number ->
        ``count`` (device uint32/int32 [1]).  Nothing is read back."""
        torch = _torch()
        if entries.dtype != torch.uint8 or entries.dim() != 2 or entries.shape[1] != _lib.TILE_ENTRY_BYTES \
                or not entries.is_contiguous() or count.numel() != 1 or count.element_size() != 4:
            raise ValueError("entries must be uint8 [capacity, TILE_ENTRY_BYTES] (contiguous), count one 32-bit word")
        self._check(self._L.rt_compact_tiles(self._h, slab.data_ptr(), entries.data_ptr(), count.data_ptr(),
                                             int(entries.shape[0])))

    def place_tiles(self, entries, counts, width, height, tile_rows, out, table=None, on_current_stream=False):
        """Root side (rt_place_tiles): gathered ``entries`` uint8 [n_ranks, capacity, TILE_ENTRY_BYTES] and ``counts``
        (device, 32-bit, [n_ranks]) -> ``out`` uint8 [F, H, W, 3], black where no entry covers a tile.
        ``on_current_stream``: launch on torch's current stream for this call only (BatchExchange places on its side
        stream, beside the renders of later batches), then go back to the stream of use_torch_stream()."""
        torch = _torch()
        if on_current_stream:
            mine = self._stream
            self.use_torch_stream()
            try:
                return self.place_tiles(entries, counts, width, height, tile_rows, out, table)
            finally:
                self._check(self._L.rt_set_stream(self._h, ctypes.c_void_p(mine), 0))
                self._stream = mine
        from .tiling import max_local_rows
        n_ranks, cap = int(entries.shape[0]), int(entries.shape[1])
        nf = int(out.shape[0])
        _check_out(out, (nf, height, width, 3), torch.uint8, self.device)
        need = n_ranks * nf * (((width + 7) // 8) | 1) * ((max_local_rows(height, tile_rows, n_ranks) + 7) // 8)
        if table is None or table.numel() < need:
            table = self._alloc((need,), torch.int32)
        if not entries.is_contiguous() or counts.element_size() != 4 or counts.numel() != n_ranks:
            raise ValueError("entries must be contiguous, counts 32-bit [n_ranks]")
        self._check(self._L.rt_place_tiles(self._h, entries.data_ptr(), counts.data_ptr(), n_ranks, cap, nf, int(width),
                                           int(height), int(tile_rows), table.data_ptr(), int(table.numel()),
                                           out.data_ptr()))
        return table

    # -- RCCL behind the C ABI (rt_comm_*: the library's own communicator) -------------
    def comm_init(self, group=None):
        """Collective over the ranks of torch.distributed's ``group``: rank 0 makes RCCL's unique id (rt_comm_unique_id),
        torch.distributed hands its 128 bytes round (plumbing: any channel would do), every rank joins
        (rt_comm_init).  After this the exchange of BatchExchange(comm=renderer) runs inside librt355."""
        import torch.distributed as dist
        torch = _torch()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # Every rank first makes an id of its own: that binds librccl and fails where it cannot be bound, BEFORE anything
        # collective is entered — ncclCommInitRank would leave the ranks that did reach it waiting for the one that did not.
        buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
        rc = self._L.rt_comm_unique_id(buf)
        err = "" if rc == 0 else (self._L.rt_last_error(None) or b"unknown error").decode()
        dev = f"cuda:{self.device}" if dist.get_backend(group) == "nccl" else "cpu"
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            raise RuntimeError("rt_comm_* is not available on every rank" + (f" (this rank: {err})" if err else ""))
        box = [bytes(buf) if rank == 0 else None]          # rank 0's id is the communicator's
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = (ctypes.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
        self._check(self._L.rt_comm_init(self._h, ident, world, rank))

    def comm_init_single(self):
        """A communicator of ONE rank (no torch.distributed needed): the one-GPU rehearsal of the calls."""
        buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
        self._check(self._L.rt_comm_unique_id(buf))
        self._check(self._L.rt_comm_init(self._h, buf, 1, 0))

    def comm_info(self) -> dict:
        n, r, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self._L.rt_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v)))
        return dict(n_ranks=n.value, rank=r.value, rccl_version=v.value)

    def comm_destroy(self):
        self._check(self._L.rt_comm_destroy(self._h))

    def comm_all_gather_counts(self, count, counts_all):
        """counts_all[r] = rank r's count (device, 32-bit), on torch's CURRENT stream (rt_comm_all_gather_counts)."""
        torch = _torch()
        if count.numel() != 1 or count.element_size() != 4 or counts_all.element_size() != 4 or not counts_all.is_contiguous():
            raise ValueError("count: one 32-bit word; counts_all: contiguous 32-bit [n_ranks]")
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._L.rt_comm_all_gather_counts(self._h, count.data_ptr(), counts_all.data_ptr(), ctypes.c_void_p(s)))

    def comm_gather(self, send, send_bytes, gathered, recv_bytes, rank_stride_bytes, root=0):
        """rt_comm_gather on torch's CURRENT stream: the first ``send_bytes`` bytes of ``send`` -> ``gathered`` (root
        only; else None) at rank * rank_stride_bytes; ``recv_bytes``: what each rank sends (read on the root)."""
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        rb = (ctypes.c_int64 * len(recv_bytes))(*[int(b) for b in recv_bytes]) if recv_bytes is not None else None
        self._check(self._L.rt_comm_gather(self._h, send.data_ptr(), int(send_bytes),
                                           gathered.data_ptr() if gathered is not None else None, rb,
                                           int(rank_stride_bytes), int(root), ctypes.c_void_p(s)))

    def ray_counts(self) -> dict:
        """{'primary','reflection','shadow','shaded_hits','total'} since the last reset (syncs)."""
        a = (ctypes.c_uint64 * 4)()
        self._check(self._L.rt_get_ray_counts(self._h, a))
        d = dict(primary=a[0], reflection=a[1], shadow=a[2], shaded_hits=a[3])
        d["total"] = a[0] + a[1] + a[2]
        return d

    def reset_ray_counts(self):
        self._check(self._L.rt_reset_ray_counts(self._h))


def _same_scene(a: PackedScene, b: PackedScene) -> bool:
    """Bitwise equality of everything upload_scene sends to the device."""
    if a.area_samples != b.area_samples:
        return False
    for name in ("geom4", "mat4", "spec", "lights", "refr", "cube_box", "cube_mat4", "cube_spec", "cube_refr", "order"):
        x, y = getattr(a, name), getattr(b, name)
        if x.shape != y.shape or x.tobytes() != y.tobytes():
            return False
    return True


def _to_host(t):
    """Device tensor -> numpy array through a pinned staging buffer of its own (torch caches pinned
    allocations; the array keeps its buffer alive): ~3x the rate of a pageable copy."""
    torch = _torch()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _dims(width, height):
    width, height = int(width), int(height)
    if width < 0 or height < 0:
        raise ValueError("negative dimensions are not allowed")       # np.zeros((h, w, 3)), :314
    return width, height


def _check_out(t, shape, dtype, device):
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() \
            or not t.is_cuda or t.device.index != device:
        raise ValueError(f"out must be a contiguous {dtype} tensor of shape {shape} on cuda:{device}")


_renderers: dict = {}


def _renderer(device=None) -> Renderer:
    torch = _torch()
    dev = torch.cuda.current_device() if device is None else int(device)
    r = _renderers.get(dev)
    if r is None:
        r = _renderers[dev] = Renderer(dev)
    r.use_torch_stream()
    return r


def render(width, height, samples=1, *, scene=None, max_depth=_DEFAULT_MAX_DEPTH, camera=None,
           rng_mode="keyed", seed=0, device=None, return_hit_index=False):
    """Drop-in for the reference's ``render(width, height, samples=1)`` (raytracer.py:310-325).

    Returns ``np.ndarray[uint8, (height, width, 3)]``, row 0 = top, ready for
    ``glTexImage2D(..., GL_RGB, GL_UNSIGNED_BYTE, image)`` (raytracer.py:346) or PIL.
    ``scene=None`` reads this module's global ``scene`` (as raytracer.py:317 does); the
    keyword-only extensions default to the reference's literals (max_depth 5, fixed pinhole).
    ``rng_mode``: the reference draws jitter / area-light samples from Python's unseeded global
    Mersenne Twister, so any stream is "the reference's"; ``"keyed"`` (default) is a per-pixel hash
    stream reproducible across GPU counts, ``"constant"`` fixes every draw at 0.5.
    """
    r = _renderer(device)
    r.upload_scene(globals()["scene"] if scene is None else scene)
    res = r.render(width, height, samples, max_depth=max_depth, camera=camera, rng_mode=rng_mode,
                   seed=seed, return_hit_index=return_hit_index)
    if return_hit_index:
        return _to_host(res[0]), _to_host(res[1])
    return _to_host(res)


def render_chunk(args, *, max_depth=_DEFAULT_MAX_DEPTH, camera=None, rng_mode="keyed", seed=0,
                 device=None):
    """Drop-in for ``render_chunk(args)`` (raytracer.py:299-308): ``args`` is the reference's tuple
    ``(x_start, x_end, y_start, y_end, width, height, scene, samples)``; returns the clamped,
    unquantised colour as ``np.ndarray[float64, (y_end-y_start, x_end-x_start, 3)]`` (computed in
    float32 on the GPU)."""
    x_start, x_end, y_start, y_end, width, height, scn, samples = args
    r = _renderer(device)
    r.upload_scene(scn)
    t = r.render_chunk(x_start, x_end, y_start, y_end, width, height, samples, max_depth=max_depth,
                       camera=camera, rng_mode=rng_mode, seed=seed)
    return t.cpu().numpy().astype(np.float64)
