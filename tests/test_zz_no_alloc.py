"""rt_render_noalloc (include/rt355.h): the entry point that allocates nothing — its kernels' tile records, work
counters, occupancy bits, frame record and primary-ray table lie in a fixed device array instead of buffers grown
with hipMalloc.  Called through the raw C ABI and compared byte for byte with rt_render, two calls in a row, the
second with a smaller frame (the array keeps the first call's contents), plus the refusals.

Collected last (file name): the round's end runs `pytest -x`, so a failure here hides no other test."""
import ctypes

import numpy as np
import pytest

from raytracer_amd import _lib
from raytracer_amd.scenes import RtCamera, default_scene, generate_scene, orbit_camera, pack_scene, refractive_scene

pytestmark = pytest.mark.gpu


def _upload(L, h, scene):
    p = pack_scene(scene)
    lights = p.ct_lights()
    g, m, s = (np.ascontiguousarray(a, np.float32) for a in (p.geom4, p.mat4, p.spec))
    assert L.rt_upload_scene(h, g.ctypes.data, m.ctypes.data, s.ctypes.data, p.n_spheres, ctypes.byref(lights)) == 0
    if p.has_refraction:
        f = np.ascontiguousarray(p.refr[:, 0], np.float32)
        ri = np.ascontiguousarray(p.refr[:, 1], np.float32)
        assert L.rt_set_refraction(h, f.ctypes.data, ri.ctypes.data, p.n_spheres) == 0


def test_no_alloc_entry_point_matches_rt_render():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.rt_create(0, ctypes.byref(h)) == 0
    # the context launches on torch's current stream, so the fills below, the renders and the comparisons are ordered
    assert L.rt_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream(0).cuda_stream), 0) == 0
    try:
        def both(w, hh, D, mode, seed, cam):
            c = RtCamera.from_camera(cam)
            a = torch.full((hh, w, 3), 9, dtype=torch.uint8, device="cuda:0")
            b = torch.full((hh, w, 3), 5, dtype=torch.uint8, device="cuda:0")
            ha = torch.full((hh, w), -7, dtype=torch.int32, device="cuda:0")
            hb = torch.full((hh, w), -9, dtype=torch.int32, device="cuda:0")
            assert L.rt_render(h, ctypes.byref(c), w, hh, 1, D, mode, seed, a.data_ptr(), ha.data_ptr()) == 0
            assert L.rt_render_noalloc(h, ctypes.byref(c), w, hh, 1, D, mode, seed, b.data_ptr(), hb.data_ptr()) == 0, \
                L.rt_last_error(h)
            assert L.rt_synchronize(h) == 0
            return a, b, ha, hb

        _upload(L, h, generate_scene(8, 1, keep_default=True))
        # two calls in a row, the second with a smaller frame; then a ragged one, another camera, the constant RNG
        for w, hh, D, mode, seed, cam in ((1920, 1080, 5, 1, 3, None), (640, 360, 5, 1, 4, None),
                                           (333, 187, 8, 1, 5, orbit_camera(40, 256)), (64, 48, 3, 0, 0, None)):
            a, b, ha, hb = both(w, hh, D, mode, seed, cam)
            assert torch.equal(a, b) and torch.equal(ha, hb), (w, hh)
            assert int(a.max()) > 9                                    # something was rendered
        # back to back on the stream without a host sync in between: the second call reuses the array in stream order
        c = RtCamera.from_camera(None)
        big = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
        small = torch.empty((270, 480, 3), dtype=torch.uint8, device="cuda:0")
        assert L.rt_render_noalloc(h, ctypes.byref(c), 1920, 1080, 1, 5, 1, 11, big.data_ptr(), None) == 0
        assert L.rt_render_noalloc(h, ctypes.byref(c), 480, 270, 1, 5, 1, 12, small.data_ptr(), None) == 0
        assert L.rt_synchronize(h) == 0
        ref_big = torch.empty_like(big); ref_small = torch.empty_like(small)
        assert L.rt_render(h, ctypes.byref(c), 1920, 1080, 1, 5, 1, 11, ref_big.data_ptr(), None) == 0
        assert L.rt_render(h, ctypes.byref(c), 480, 270, 1, 5, 1, 12, ref_small.data_ptr(), None) == 0
        assert L.rt_synchronize(h) == 0
        assert torch.equal(big, ref_big) and torch.equal(small, ref_small)
        # a larger scene (64 spheres: the wave-culled visibility pass, a larger primary-ray table) at 4K: fits
        _upload(L, h, generate_scene(64, 2))
        a, b, ha, hb = both(3840, 2160, 5, 1, 1, None)
        assert torch.equal(a, b) and torch.equal(ha, hb)
        # refusals: a frame whose records do not fit, samples > 1, a refractive scene — each with a message
        out = torch.empty(16, dtype=torch.uint8, device="cuda:0")    # never written: the calls fail before any launch
        assert L.rt_render_noalloc(h, ctypes.byref(c), 7680, 4320, 1, 5, 1, 0, out.data_ptr(), None) == -1
        assert b"scratch" in L.rt_last_error(h)
        assert L.rt_render_noalloc(h, ctypes.byref(c), 64, 48, 2, 5, 1, 0, out.data_ptr(), None) == -1
        assert b"samples" in L.rt_last_error(h)
        _upload(L, h, refractive_scene())
        assert L.rt_render_noalloc(h, ctypes.byref(c), 64, 48, 1, 5, 1, 0, out.data_ptr(), None) == -1
        assert b"refracti" in L.rt_last_error(h)
        assert L.rt_render_noalloc(h, ctypes.byref(c), 64, 48, 1, 5, 1, 0, None, None) == -1
    finally:
        L.rt_destroy(h)


def test_no_alloc_entry_point_allocates_nothing():
    """A fresh context whose FIRST render call is rt_render_noalloc (nothing has grown the context's own buffers, so a
    call that used them instead of the scratch array would fault or fail), with the device's free memory unchanged
    across the calls; the frame equals rt_render's on another context."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    L = _lib.load()
    h, h2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.rt_create(0, ctypes.byref(h)) == 0 and L.rt_create(0, ctypes.byref(h2)) == 0
    stream = ctypes.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    assert L.rt_set_stream(h, stream, 0) == 0 and L.rt_set_stream(h2, stream, 0) == 0
    try:
        scene = generate_scene(8, 1, keep_default=True)
        _upload(L, h, scene)
        _upload(L, h2, scene)
        c = RtCamera.from_camera(None)
        out = torch.zeros((2, 540, 960, 3), dtype=torch.uint8, device="cuda:0")
        ref = torch.zeros((540, 960, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        for i in range(2):
            assert L.rt_render_noalloc(h, ctypes.byref(c), 960, 540, 1, 5, 1, 7, out[i].data_ptr(), None) == 0, L.rt_last_error(h)
        assert L.rt_synchronize(h) == 0
        free1, _ = torch.cuda.mem_get_info(0)
        assert free1 == free0, (free0, free1)
        assert L.rt_render(h2, ctypes.byref(c), 960, 540, 1, 5, 1, 7, ref.data_ptr(), None) == 0
        assert L.rt_synchronize(h2) == 0
        assert torch.equal(out[0], ref) and torch.equal(out[1], ref) and int(ref.max()) > 9
    finally:
        L.rt_destroy(h)
        L.rt_destroy(h2)
