"""Size-independent properties of the HIP path at BASELINE's full sizes, API mirror behaviour and
error handling (all through the C ABI)."""
import ctypes

import numpy as np
import pytest

import raytracer_amd
from raytracer_amd import _lib, tiling
from raytracer_amd.scenes import Cube, RtCamera, Vector3, default_scene, generate_scene, orbit_camera

pytestmark = pytest.mark.gpu


def test_row_tiles_reassemble_byte_identical(renderer):
    """Config 4's sharding on one GPU: render each rank's tiles separately, scatter, compare."""
    import torch
    scene = generate_scene(64, 2)
    renderer.upload_scene(scene)
    for (w, h, tr, G) in ((1920, 1080, 8, 2), (1920, 1080, 8, 8), (3840, 2160, 16, 4), (333, 187, 8, 3)):
        full = renderer.render(w, h, 1, max_depth=5, seed=5)
        rows_max = tiling.max_local_rows(h, tr, G)
        gathered = torch.zeros((G, rows_max * w * 3), dtype=torch.uint8, device="cuda:0")
        for r in range(G):
            loc = renderer.render_rows(w, h, 1, tile_rows=tr, tile_first=r, tile_stride=G, max_depth=5, seed=5)
            assert loc.shape[0] == tiling.local_row_count(h, tr, r, G)
            gathered[r, :loc.numel()] = loc.reshape(-1)
            rows = torch.from_numpy(tiling.local_rows(h, tr, r, G)).cuda()
            assert torch.equal(loc, full[rows])
        img = renderer.scatter_rows(gathered, G, w, h, tr)
        assert torch.equal(img, full)


def test_deterministic_and_seed_sensitive(renderer):
    import torch
    renderer.upload_scene(generate_scene(8, 1, keep_default=True))
    a = renderer.render(1920, 1080, 1, seed=1)
    b = renderer.render(1920, 1080, 1, seed=1)
    c = renderer.render(1920, 1080, 1, seed=2)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_mode_c_is_independent_of_sample_count(renderer):
    """In mode C every sample is the same ray, and (k*c)*(1/k) is exact for k a power of two."""
    import torch
    renderer.upload_scene(default_scene())
    a = renderer.render(640, 480, 1, rng_mode="C")
    b = renderer.render(640, 480, 4, rng_mode="C")
    assert torch.equal(a, b)


def test_chunk_matches_frame(renderer):
    """render_chunk tiles quantise to the same bytes as render (raytracer.py:305 vs :318,:325)."""
    renderer.upload_scene(generate_scene(8, 1, keep_default=True))
    full = renderer.render(1920, 1080, 1, seed=9).cpu().numpy()
    for (x0, x1, y0, y1) in ((0, 1920, 0, 1080), (100, 421, 333, 400), (1900, 1920, 1070, 1080), (5, 6, 7, 8)):
        c = renderer.render_chunk(x0, x1, y0, y1, 1920, 1080, 1, seed=9).cpu().numpy()
        assert c.min() >= 0 and c.max() <= 1
        assert np.array_equal((c * np.float32(255)).astype(np.uint8), full[y0:y1, x0:x1])


def test_frames_batch_equals_single_renders(renderer):
    import torch
    renderer.upload_scene(generate_scene(8, 1, keep_default=True))
    cams = [orbit_camera(f) for f in (0, 3, 77)]
    batch = renderer.render_frames(cams, [0, 3, 77], 640, 360, 1)
    for i, f in enumerate((0, 3, 77)):
        assert torch.equal(batch[i], renderer.render(640, 360, 1, camera=cams[i], seed=f))
    # more frames than one launch holds (RT_FRAMES_PER_LAUNCH = 1024): two launches, same frames
    n = _lib.FRAMES_PER_LAUNCH + 6
    cams = [orbit_camera(f) for f in range(n)]
    batch = renderer.render_frames(cams, list(range(n)), 160, 90, 1)
    for f in (0, 63, 64, 255, 256, 700, 1023, 1024, n - 1):     # (frame numbers travel in 13 bits of the queue records)
        assert torch.equal(batch[f], renderer.render(160, 90, 1, camera=cams[f], seed=f))
    # refractive scene (float pixel sums, resolved per launch): frames beyond 255 of one launch
    from raytracer_amd.scenes import refractive_scene
    renderer.upload_scene(refractive_scene())
    batch = renderer.render_frames(cams[:300], list(range(300)), 160, 90, 1)
    for f in (0, 255, 256, 299):
        d = (batch[f].to(torch.int16) - renderer.render(160, 90, 1, camera=cams[f], seed=f).to(torch.int16)).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01


def test_8k_frame_runs_and_counts(renderer):
    """Config 4 full size on one GPU: finishes, is not black, counters obey the closed form."""
    renderer.upload_scene(generate_scene(1024, 3, 7680, 4320))
    renderer.reset_ray_counts()
    img = renderer.render(7680, 4320, 1, max_depth=5, seed=0)
    rc = renderer.ray_counts()
    assert rc["primary"] == 7680 * 4320 and rc["shadow"] == 13 * rc["shaded_hits"]
    assert rc["shaded_hits"] > 0.2 * rc["primary"]
    assert int(img.max()) > 100


def test_module_level_render_mirrors_reference_api(renderer):
    """render(width, height, samples) -> uint8 [H,W,3], reading the module-global scene."""
    img = raytracer_amd.render(64, 48)
    assert isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.shape == (48, 64, 3)
    assert img.any()
    img2, hit = raytracer_amd.render(64, 48, 1, return_hit_index=True)
    assert np.array_equal(img, img2) and hit.shape == (48, 64) and hit.max() <= 2
    saved = raytracer_amd.scene
    try:
        s = default_scene(); s["spheres"] = []
        raytracer_amd.set_scene(s)
        assert not raytracer_amd.render(16, 8).any()
    finally:
        raytracer_amd.set_scene(saved)
    ch = raytracer_amd.render_chunk((8, 24, 4, 20, 64, 48, default_scene(), 1))
    assert ch.dtype == np.float64 and ch.shape == (16, 16, 3)
    assert np.array_equal((ch.astype(np.float32) * np.float32(255)).astype(np.uint8), img[4:20, 8:24])
    assert raytracer_amd.render(0, 0).shape == (0, 0, 3)
    with pytest.raises(ValueError):
        raytracer_amd.render(-1, 4)


def test_abi_error_paths():
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.rt_create(0, ctypes.byref(h)) == 0
    try:
        cam = RtCamera.from_camera(None)
        import torch
        buf = torch.empty(16 * 16 * 3, dtype=torch.uint8, device="cuda:0")
        # render before upload
        assert L.rt_render(h, ctypes.byref(cam), 16, 16, 1, 5, 1, 0, buf.data_ptr(), None) == -3
        assert b"rt_upload_scene" in L.rt_last_error(h)
        r = raytracer_amd.Renderer(0)
        p = r.upload_scene(default_scene())
        lights = p.ct_lights()
        g = np.ascontiguousarray(p.geom4); m = np.ascontiguousarray(p.mat4); s = np.ascontiguousarray(p.spec)
        up = lambda n, L_=lights, g_=g: L.rt_upload_scene(h, g_.ctypes.data, m.ctypes.data, s.ctypes.data, n, ctypes.byref(L_))
        assert up(3) == 0
        assert up(-1) == -1 and up(5000) == -1
        bad = np.array(g); bad[0, 0] = np.nan
        assert up(3, lights, bad) == -1 and b"non-finite" in L.rt_last_error(h)
        bl = p.ct_lights(); bl.area_samples = 0
        assert up(3, bl) == -1
        assert up(3) == 0
        for args in ((0, 16, 1, 5, 1), (16, 16, 0, 5, 1), (16, 16, 1, -1, 1), (16, 16, 1, 65, 1), (16, 16, 1, 5, 7)):
            w, hh, spp, D, mode = args
            assert L.rt_render(h, ctypes.byref(cam), w, hh, spp, D, mode, 0, buf.data_ptr(), None) == -1, args
        assert L.rt_render(h, ctypes.byref(cam), 16, 16, 1, 5, 1, 0, None, None) == -1
        assert L.rt_render(h, None, 16, 16, 1, 5, 1, 0, buf.data_ptr(), None) == -1
        n = ctypes.c_int(-1)
        assert L.rt_render_rows(h, ctypes.byref(cam), 16, 16, 1, 5, 1, 0, 8, 2, 2, buf.data_ptr(), None, ctypes.byref(n)) == -1
        assert L.rt_render_rows(h, ctypes.byref(cam), 16, 16, 1, 5, 1, 0, 8, 3, 4, buf.data_ptr(), None, ctypes.byref(n)) == 0
        assert n.value == 0                                  # rank 3 of 4 owns no tile of a 16-row frame
        assert L.rt_render(h, ctypes.byref(cam), 16, 16, 1, 5, 1, 0, buf.data_ptr(), None) == 0
        assert L.rt_synchronize(h) == 0
        r.close()
    finally:
        L.rt_destroy(h)


def test_refraction_abi(renderer):
    """rt_set_refraction: argument checks, reset by rt_upload_scene, RT_KERNEL_PIXEL refused."""
    L = _lib.load()
    from raytracer_amd.scenes import refractive_scene
    pk = renderer.upload_scene(refractive_scene())
    n = pk.n_spheres
    f = np.ascontiguousarray(pk.refr[:, 0]); ri = np.ascontiguousarray(pk.refr[:, 1])
    assert L.rt_set_refraction(renderer._h, f.ctypes.data, ri.ctypes.data, n + 1) == -1
    assert L.rt_set_refraction(renderer._h, None, ri.ctypes.data, n) == -1
    bad = ri.copy(); bad[0] = 0.0
    assert L.rt_set_refraction(renderer._h, f.ctypes.data, bad.ctypes.data, n) == -1
    assert b"refractive_index" in L.rt_last_error(renderer._h)
    renderer.set_kernel("pixel")
    try:
        with pytest.raises(_lib.RtError):
            renderer.render(64, 48, 1)
    finally:
        renderer.set_kernel("stream")
    a = renderer.render(64, 48, 1, rng_mode="C").cpu().numpy()
    renderer.upload_scene(default_scene())          # resets refraction to Material's defaults
    renderer.set_kernel("pixel")
    try:
        b = renderer.render(64, 48, 1, rng_mode="C").cpu().numpy()
    finally:
        renderer.set_kernel("stream")
    assert not np.array_equal(a, b)


def test_cubes_abi(renderer):
    """rt_set_cubes: argument checks, reset by rt_upload_scene, RT_KERNEL_PIXEL refused, hit indices
    reported as positions in the scene's object list, batches and row tiles agree with single frames."""
    import torch
    from raytracer_amd import tiling
    from raytracer_amd.scenes import cube_scene
    L = _lib.load()
    pk = renderer.upload_scene(cube_scene(True))
    assert pk.n_cubes == 3 and list(pk.order) == [1, 2, 4, 0, 3, 5]
    box, mat, spec = (np.ascontiguousarray(a) for a in (pk.cube_box, pk.cube_mat4, pk.cube_spec))
    f = np.ascontiguousarray(pk.cube_refr[:, 0]); ri = np.ascontiguousarray(pk.cube_refr[:, 1])
    h = renderer._h
    assert L.rt_set_cubes(h, box.ctypes.data, mat.ctypes.data, spec.ctypes.data, f.ctypes.data, ri.ctypes.data, -1) == -1
    assert L.rt_set_cubes(h, box.ctypes.data, mat.ctypes.data, spec.ctypes.data, f.ctypes.data, ri.ctypes.data, 1100) == -1
    assert L.rt_set_cubes(h, None, mat.ctypes.data, spec.ctypes.data, None, None, 3) == -1
    bad = box.copy(); bad[1, 2] = np.inf
    assert L.rt_set_cubes(h, bad.ctypes.data, mat.ctypes.data, spec.ctypes.data, None, None, 3) == -1
    assert b"cube 1" in L.rt_last_error(h)
    badri = ri.copy(); badri[0] = -1.0
    assert L.rt_set_cubes(h, box.ctypes.data, mat.ctypes.data, spec.ctypes.data, f.ctypes.data, badri.ctypes.data, 3) == -1
    assert L.rt_set_cubes(h, box.ctypes.data, mat.ctypes.data, spec.ctypes.data, f.ctypes.data, ri.ctypes.data, 3) == 0
    renderer.set_kernel("pixel")
    try:
        with pytest.raises(_lib.RtError):
            renderer.render(64, 48, 1)
    finally:
        renderer.set_kernel("stream")
    a, hit = renderer.render(200, 120, 1, rng_mode="K", seed=3, return_hit_index=True)
    assert set(np.unique(hit.cpu().numpy())) == {-1, 0, 1, 2, 4, 5}     # list positions; cube 3 is out of view
    # opaque cubes (NULL refraction arrays = Material's defaults): the fast any-hit path; batches and tiles
    assert L.rt_set_cubes(h, box.ctypes.data, mat.ctypes.data, spec.ctypes.data, None, None, 3) == 0
    one = renderer.render(200, 120, 1, rng_mode="K", seed=3)
    assert not torch.equal(one, a)
    cams = [None, orbit_camera(3, 64), None]
    batch = renderer.render_frames(cams, [3, 9, 3], 200, 120, 1, rng_mode="K")
    assert torch.equal(batch[0], one) and torch.equal(batch[2], one) and not torch.equal(batch[1], one)
    rows = renderer.render_rows(200, 120, 1, tile_rows=8, tile_first=1, tile_stride=2, rng_mode="K", seed=3)
    assert torch.equal(rows, one[torch.as_tensor(tiling.local_rows(120, 8, 1, 2), device=one.device)])
    renderer.upload_scene(default_scene())          # the object list holds spheres only again
    renderer.set_kernel("pixel")
    try:
        b = renderer.render(200, 120, 1, rng_mode="K", seed=3)
    finally:
        renderer.set_kernel("stream")
    assert torch.equal(b, renderer.render(200, 120, 1, rng_mode="K", seed=3)) and not torch.equal(b, one)


def test_occluder_table_changes_no_byte(renderer):
    """The shadow scan restricted to rt_occluder_table's candidates renders the same bytes and hit maps as
    the scan over every sphere (what the reference does), from 3 to 1024 spheres, with cubes, other
    cameras and light sample counts."""
    import torch
    from raytracer_amd.scenes import cube_scene
    cases = [(default_scene(), 640, 480, 5), (generate_scene(8, 1, keep_default=True), 1920, 1080, 5),
             (generate_scene(64, 2), 1920, 1080, 5), (generate_scene(64, 2), 960, 540, 8),
             (generate_scene(200, 5), 640, 360, 5), (generate_scene(1024, 3), 960, 540, 5), (cube_scene(), 640, 360, 5)]
    cases[3][0]["area_light"].samples = 20
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (True, False):
                renderer.set_occluder_table(on)
                renderer.upload_scene(scene)
                out.append([renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=9, camera=cam, return_hit_index=True)
                            for cam in (None, orbit_camera(37, 256), orbit_camera(150, 256))])
            for (a, ha), (b, hb) in zip(*out):
                assert torch.equal(a, b) and torch.equal(ha, hb)
    finally:
        renderer.set_occluder_table(True)


def test_reflect_lists_change_no_byte(renderer):
    """Scenes with 256 spheres or more: reflection rays through the direction table in its list form find the same
    hits as the plain scan (RT_OPT_REFLECT_TABLE off): byte-identical frames and hit
    maps, depth up to 8, other cameras, a scene with two huge spheres (they reach every direction of their
    neighbours and list every sphere themselves), 2000 spheres."""
    import torch
    from raytracer_amd.scenes import Material, Sphere, Vector3
    big = generate_scene(300, 11)
    big["spheres"].append(Sphere(Vector3(0, -1004, -10), 1000, Material(Vector3(0.8, 0.8, 0.8), specular=0.3, reflection=0.4)))
    big["spheres"].insert(5, Sphere(Vector3(30, 0, -40), 25, Material(Vector3(0.9, 0.5, 0.5), specular=0.3, reflection=0.6)))
    cases = [(generate_scene(256, 4), 640, 360, 5), (generate_scene(400, 5), 640, 360, 8), (generate_scene(1024, 3), 960, 540, 5),
             (generate_scene(2000, 6), 480, 270, 3), (big, 640, 360, 6)]
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (True, False):
                renderer.set_reflect_table(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                frames = [renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=3, camera=cam, return_hit_index=True)
                          for cam in (None, orbit_camera(37, 256), orbit_camera(150, 256, radius=14.0))]
                out.append((frames, renderer.ray_counts()))
            (fa, ca), (fb, cb) = out
            assert ca == cb and ca["reflection"] > 0
            for (a, ha), (b, hb) in zip(fa, fb):
                assert torch.equal(a, b) and torch.equal(ha, hb)
    finally:
        renderer.set_reflect_table(True)


def test_reflect_table_changes_no_byte(renderer):
    """Reflection rays through the per-object direction table (sphere-only opaque scenes with 4..255 spheres) find
    the same hits as the plain scan: byte-identical frames and hit maps with RT_OPT_REFLECT_TABLE on and off, depth
    up to 8, other cameras, tables of every bucket count (16 / 8 / 4 per face edge, chosen by size), spheres that
    nearly touch, and a scene with a huge floor sphere that the others rest on."""
    import torch
    from raytracer_amd.scenes import Material, Sphere, Vector3
    floor = generate_scene(40, 11)
    floor["spheres"].append(Sphere(Vector3(0, -1003, -12), 1000, Material(Vector3(0.8, 0.8, 0.8), specular=0.3, reflection=0.5)))
    pair = generate_scene(20, 13)
    pair["spheres"].append(Sphere(Vector3(-0.6, 0.2, -4), 0.6, Material(Vector3(0.9, 0.3, 0.3), specular=1.0, reflection=0.7)))
    pair["spheres"].append(Sphere(Vector3(0.6005, 0.2, -4), 0.6, Material(Vector3(0.3, 0.3, 0.9), specular=1.0, reflection=0.7)))
    cases = [(generate_scene(8, 1, keep_default=True), 640, 360, 5), (generate_scene(16, 4), 640, 360, 5), (generate_scene(64, 2), 960, 540, 8), (generate_scene(100, 5), 640, 360, 5),
             (generate_scene(255, 6), 640, 360, 6), (floor, 640, 360, 6), (pair, 640, 360, 8)]
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (True, False):
                renderer.set_reflect_table(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                frames = [renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=3, camera=cam, return_hit_index=True)
                          for cam in (None, orbit_camera(37, 256), orbit_camera(150, 256, radius=14.0))]
                out.append((frames, renderer.ray_counts()))
            (fa, ca), (fb, cb) = out
            assert ca == cb and ca["reflection"] > 0
            for (a, ha), (b, hb) in zip(fa, fb):
                assert torch.equal(a, b) and torch.equal(ha, hb)
    finally:
        renderer.set_reflect_table(True)


def test_reflect_table_on_refractive_scenes(renderer):
    """Refractive scenes send their reflected AND refracted rays (origins 0.001 outside / inside the object) through
    the same direction table.  Pixel sums are float atomics there, so frames agree up to summation order; the tree is
    the same: identical ray counts and hit maps.  The glass-sphere scene, and 40 generated spheres every third of
    glass (some of them seen through others), depth 5 and 8."""
    import torch
    from raytracer_amd.scenes import refractive_scene
    many = generate_scene(40, 12)
    for i, s_ in enumerate(many["spheres"]):
        if i % 3 == 0:
            s_.material.refraction, s_.material.refractive_index = 0.6, 1.4
    cases = [(refractive_scene(), 640, 360, 5), (many, 640, 360, 5), (many, 480, 270, 8)]
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (False, True):
                renderer.set_reflect_table(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                img, hit = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=5, return_hit_index=True)
                out.append((img.to(torch.int16), hit, renderer.ray_counts()))
            (a, ha, ca), (b, hb, cb) = out
            assert ca["reflection"] > 0 and ca == cb and torch.equal(ha, hb)
            d = (a - b).abs()
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    finally:
        renderer.set_reflect_table(True)


def test_queue_spill_changes_only_summation_order(renderer):
    """Refractive scenes: a wave whose work queue is full sets records aside in device memory
    (RT_OPT_QUEUE_SPILL, default) or only shrinks its passes.  Same tree nodes either way: identical ray counts and
    hit maps, and images that differ by float summation order only (<= 1/255 on a few pixels).  Glass cubes
    at depth 5, 9 and 14 (spill stack 12, 20 and 30 blocks deep at most), the glass sphere scene, 2 samples."""
    import torch
    from raytracer_amd.scenes import cube_scene, refractive_scene
    cases = [(cube_scene(glass=True), 480, 270, 5, 1), (cube_scene(glass=True), 240, 136, 9, 1),
             (cube_scene(glass=True), 96, 54, 14, 1), (refractive_scene(), 640, 360, 5, 2)]
    try:
        for scene, w, h, D, spp in cases:
            out = []
            for on in (False, True, 2, 3):     # 2, 3: a stack that fills up, so both rules take turns
                renderer.set_queue_spill(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                img, hit = renderer.render(w, h, spp, max_depth=D, rng_mode="K", seed=5, return_hit_index=True)
                out.append((img.to(torch.int16), hit, renderer.ray_counts()))
            a, ha, ca = out[0]
            assert ca["total"] > w * h
            for b, hb, cb in out[1:]:
                assert torch.equal(ha, hb) and ca == cb
                d = (a - b).abs()
                assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    finally:
        renderer.set_queue_spill(True)


def test_tile_flush_changes_no_byte(renderer):
    """The stream kernel's coalesced output path (finished pixels staged per tile in LDS, whole tiles stored:
    RT_OPT_TILE_FLUSH, default) writes the bytes the pixel-by-pixel stores write: full frames, ragged sizes
    (partial edge tiles, widths that are no multiple of 4: the byte path), row tiles whose 8-row tiles straddle,
    deep bounces (tiles that stay in their slot for many passes), frame batches with a padded frame stride,
    cubes, many small spheres (sparse tiles: slots run out), both RNG modes."""
    import torch
    from raytracer_amd.scenes import cube_scene
    cases = [(generate_scene(8, 1, keep_default=True), 1920, 1080, 5), (generate_scene(64, 2), 1920, 1080, 8),
             (default_scene(), 333, 187, 5), (generate_scene(64, 2), 1001, 77, 5), (generate_scene(1024, 3), 960, 540, 5),
             (generate_scene(300, 7), 640, 360, 5), (cube_scene(), 640, 357, 5), (default_scene(), 5, 3, 5)]
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (True, False):
                renderer.set_tile_flush(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                frames = [renderer.render(w, h, 1, max_depth=D, rng_mode=mode, seed=9, camera=cam, return_hit_index=True)
                          for cam, mode in ((None, "K"), (orbit_camera(37, 256), "K"), (None, "C"))]
                rows = renderer.render_rows(w, h, 1, tile_rows=5, tile_first=1, tile_stride=3, max_depth=D, seed=2)
                F = 5
                slab = torch.full((F, h + 3, w, 3), 7, dtype=torch.uint8, device="cuda:0")
                renderer.render_frames([orbit_camera(f, 64) for f in range(F)], list(range(F)), w, h, 1, max_depth=D, out=slab)
                out.append((frames, rows, slab, renderer.ray_counts()))
            (fa, ra, sa, ca), (fb, rb, sb, cb) = out
            for (a, ha), (b, hb) in zip(fa, fb):
                assert torch.equal(a, b) and torch.equal(ha, hb)
            assert torch.equal(ra, rb) and torch.equal(sa, sb) and ca == cb
            assert int(sa[:, h:].min()) == 7 and int(sa[:, h:].max()) == 7      # rows past the frame stay untouched
    finally:
        renderer.set_tile_flush(True)


def test_samples_in_one_launch_change_no_byte(renderer):
    """2..64 samples per pixel are traced in ONE launch pair, each tile slot summing its samples in LDS in sample
    order (render_pixel :292-297); with RT_OPT_TILE_FLUSH off — and beyond 64 samples — every sample is a launch pair
    of its own that adds into a float frame in memory.  Same sums, same bytes, same ray counts: ragged sizes, deep
    bounces, cubes, sparse tiles (samples of a tile that hit nothing: no record), both RNG modes, frame batches."""
    import torch
    from raytracer_amd.scenes import cube_scene
    cases = [(generate_scene(8, 1, keep_default=True), 960, 540, 5, (2, 4)), (generate_scene(64, 2), 640, 360, 8, (3,)),
             (default_scene(), 333, 187, 5, (2, 7)), (generate_scene(300, 7), 320, 180, 5, (2,)),
             (cube_scene(), 320, 179, 5, (2,)), (default_scene(), 64, 48, 3, (64, 65)), (default_scene(), 5, 3, 5, (2,))]
    # (up to 36 spheres — a cube counts as two — the stream kernel runs its three-slot variant at four workgroups per
    # CU, above that the six-slot one at three: 30 spheres and 5 cubes take the latter's cube variant)
    many = generate_scene(30, 5)
    box_m = cube_scene()["spheres"][-1].material
    many["spheres"] = list(many["spheres"]) + [Cube(Vector3(-3.0 + 1.5 * i, -1.0, -7.0 - i), Vector3(-2.4 + 1.5 * i, -0.3, -6.2 - i), box_m)
                                               for i in range(5)]
    cases.append((many, 322, 181, 4, (2, 3)))
    try:
        for scene, w, h, D, spps in cases:
            for spp in spps:
                out = []
                for on in (True, False):
                    renderer.set_tile_flush(on)
                    renderer.upload_scene(scene)
                    renderer.reset_ray_counts()
                    frames = [renderer.render(w, h, spp, max_depth=D, rng_mode=mode, seed=9, camera=cam, return_hit_index=True)
                              for cam, mode in ((None, "K"), (orbit_camera(37, 256), "K"), (None, "C"))]
                    rows = renderer.render_rows(w, h, spp, tile_rows=5, tile_first=1, tile_stride=3, max_depth=D, seed=2)
                    F = 3
                    slab = torch.full((F, h + 2, w, 3), 7, dtype=torch.uint8, device="cuda:0")
                    renderer.render_frames([orbit_camera(f, 64) for f in range(F)], list(range(F)), w, h, spp, max_depth=D, out=slab)
                    out.append((frames, rows, slab, renderer.ray_counts()))
                (fa, ra, sa, ca), (fb, rb, sb, cb) = out
                for (a, ha), (b, hb) in zip(fa, fb):
                    assert torch.equal(a, b) and torch.equal(ha, hb), (w, h, spp)
                assert torch.equal(ra, rb) and torch.equal(sa, sb) and ca == cb, (w, h, spp)
                assert int(sa[:, h:].min()) == 7 and int(sa[:, h:].max()) == 7
    finally:
        renderer.set_tile_flush(True)


def test_stream_and_pixel_kernels_agree(renderer):
    """The work-queue kernel reorders work, never arithmetic: byte-identical to the one-thread-per-
    pixel kernel, for 1 and 3 samples, ragged sizes, both RNG modes — also at heights of 4, 8 and 16 region rows of 48 pixels,
    where the visibility pass rotates the region rows over the XCDs."""
    import torch
    for scene, (w, h), D in ((generate_scene(64, 2), (1920, 1080), 5), (generate_scene(8, 1, keep_default=True), (333, 187), 8),
                             (default_scene(), (64, 33), 3), (generate_scene(8, 1, keep_default=True), (130, 384), 5),
                             (generate_scene(150, 7), (70, 768), 4)):
        renderer.upload_scene(scene)
        for spp, mode in ((1, "K"), (3, "K"), (2, "C")):
            renderer.set_kernel("pixel")
            a, ha = renderer.render(w, h, spp, max_depth=D, rng_mode=mode, seed=4, return_hit_index=True)
            renderer.reset_ray_counts(); renderer.render(w, h, spp, max_depth=D, rng_mode=mode, seed=4)
            ca = renderer.ray_counts()
            renderer.set_kernel("stream")
            b, hb = renderer.render(w, h, spp, max_depth=D, rng_mode=mode, seed=4, return_hit_index=True)
            renderer.reset_ray_counts(); renderer.render(w, h, spp, max_depth=D, rng_mode=mode, seed=4)
            cb = renderer.ray_counts()
            assert torch.equal(a, b) and torch.equal(ha, hb) and ca == cb


def test_batched_row_tiles_with_frame_stride(renderer):
    """rt_render_frames_rows into equally sized slabs (what each rank hands the gather)."""
    import torch
    renderer.upload_scene(generate_scene(8, 1, keep_default=True))
    w, h, tr, G, F = 640, 357, 8, 3, 18                      # 18 frames -> two launches (16 + 2)
    cams = [orbit_camera(f) for f in range(F)]
    full = renderer.render_frames(cams, list(range(F)), w, h, 1)
    rows_max = tiling.max_local_rows(h, tr, G)
    gathered = torch.zeros((G, F, rows_max * w * 3), dtype=torch.uint8, device="cuda:0")
    for r in range(G):
        out = gathered[r].view(F, rows_max, w, 3)
        loc = renderer.render_frames(cams, list(range(F)), w, h, 1, out=out, tile_rows=tr, tile_first=r, tile_stride=G)
        assert loc.shape[1] == tiling.local_row_count(h, tr, r, G)
    for f in range(F):
        img = renderer.scatter_rows(gathered[:, f], G, w, h, tr)
        assert torch.equal(img, full[f])
        assert torch.equal(img, renderer.render(w, h, 1, camera=cams[f], seed=f))
    # the whole batch in one launch (rt_scatter_rows_frames), also with an odd width (byte path)
    assert torch.equal(renderer.scatter_rows_frames(gathered, G, w, h, tr), full)
    w2 = 333
    full2 = renderer.render_frames(cams[:5], list(range(5)), w2, h, 1)
    g2 = torch.zeros((G, 5, rows_max * w2 * 3), dtype=torch.uint8, device="cuda:0")
    for r in range(G):
        renderer.render_frames(cams[:5], list(range(5)), w2, h, 1, out=g2[r].view(5, rows_max, w2, 3),
                               tile_rows=tr, tile_first=r, tile_stride=G)
    assert torch.equal(renderer.scatter_rows_frames(g2, G, w2, h, tr), full2)



def test_compact_and_place_tiles_match_numpy(renderer):
    """rt_compact_tiles / rt_place_tiles (the sparse multi-GPU exchange) against the numpy restatement of their
    data format: every non-black tile of a rank's slab is among the entries, bit for bit (the occupancy bits may
    add tiles that were hit but are black), and the root assembles the frames of a whole-frame render.  Ranks are
    played one after another on one GPU; ragged sizes, widths that are no multiple of 4 (byte paths), row tiles
    of 5 rows (a rank's 8x8 tiles straddle image row tiles), several samples (occupancy bits of all samples), and
    the path without occupancy bits (samples as separate launches: tiles are found by looking at the bytes)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tile_exchange_ref as ref
    import torch
    from raytracer_amd._lib import TILE_ENTRY_BYTES
    renderer.upload_scene(generate_scene(8, 1, keep_default=True))
    F = 3
    cams = [orbit_camera(5 * f, 64) for f in range(F)]
    try:
        for (w, h, tr, G, spp, flush) in ((320, 187, 8, 3, 1, True), (333, 90, 8, 2, 1, True), (64, 48, 5, 3, 1, True),
                                          (200, 111, 16, 4, 1, True), (320, 187, 8, 1, 1, True), (160, 97, 8, 2, 3, True),
                                          (160, 97, 8, 2, 3, False), (37, 23, 8, 2, 1, False)):
            renderer.set_tile_flush(flush)
            rows_max = tiling.max_local_rows(h, tr, G)
            cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
            ent, cnt = [], []
            for r in range(G):
                slab = torch.zeros((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
                renderer.render_frames(cams, range(F), w, h, spp, out=slab, tile_rows=tr, tile_first=r, tile_stride=G)
                entries = torch.full((cap, TILE_ENTRY_BYTES), 0xAB, dtype=torch.uint8, device="cuda:0")
                count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
                renderer.compact_tiles(slab, entries, count)
                n = int(count.item())
                assert 0 < n <= cap
                n_rows = tiling.local_row_count(h, tr, r, G)
                exp, n_exp = ref.compact(slab.cpu().numpy(), n_rows, w)
                got = entries[:n].cpu().numpy()
                ids = got[:, :4].copy().view(np.uint32)[:, 0]
                assert len(set(ids.tolist())) == n                        # every tile once
                by_id = {int(i): got[k] for k, i in enumerate(ids)}
                for e in exp:
                    i = int(e[:4].copy().view(np.uint32)[0])
                    assert i in by_id and np.array_equal(by_id.pop(i), e), (w, h, r, i)
                assert all(not v[4:].any() for v in by_id.values())       # extras: hit but black
                assert n >= n_exp
                ent.append(entries); cnt.append(n)
            nmax = max(cnt)
            gathered = torch.stack([e[:nmax] for e in ent]).contiguous()
            counts = torch.tensor(cnt, dtype=torch.int32, device="cuda:0")
            out = torch.full((F, h, w, 3), 77, dtype=torch.uint8, device="cuda:0")
            renderer.place_tiles(gathered, counts, w, h, tr, out)
            full = renderer.render_frames(cams, range(F), w, h, spp)
            assert torch.equal(out, full), (w, h, tr, G, spp, flush)
            assert np.array_equal(ref.place(gathered.cpu().numpy(), cnt, F, w, h, tr), full.cpu().numpy())
        # a slab that is not the last render's frame buffer is refused
        other = torch.zeros((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
        with pytest.raises(_lib.RtError, match="last render call"):
            renderer.compact_tiles(other, entries, count)
    finally:
        renderer.set_tile_flush(True)


def test_failed_upload_never_leaves_a_stale_scene_cache(renderer):
    """A scene upload that fails midway (here: rt_set_refraction rejects a non-positive refractive index after
    rt_upload_scene has already replaced the spheres) must not leave the renderer believing that the device still
    holds the previous scene: the next render of that scene uploads it again and matches a fresh renderer."""
    import torch
    from raytracer_amd.scenes import refractive_scene
    good = generate_scene(8, 1, keep_default=True)
    renderer.upload_scene(good)
    ref = renderer.render(160, 90, 1, seed=3)
    from raytracer_amd.scenes import pack_scene
    bad = pack_scene(refractive_scene())
    bad.refr[:, 1] = -1.0                             # (pack_scene itself would refuse this; the library does too, late)
    with pytest.raises(_lib.RtError):
        renderer.upload_scene(bad)
    assert renderer._packed is None
    renderer.upload_scene(good)                       # (what the module-level render() does before every frame)
    assert torch.equal(renderer.render(160, 90, 1, seed=3), ref)


def test_occluder_table_with_refraction(renderer):
    """Refractive scenes: shadow rays look for their NEAREST hit (blocked only if it is opaque, raytracer.py:187,
    :202) among the occluder table's candidates instead of all objects.  Same tree nodes with the table on and
    off: identical ray counts and hit maps, images equal up to float summation order."""
    import torch
    from raytracer_amd.scenes import Material, Sphere, Vector3, cube_scene, refractive_scene
    many = generate_scene(40, 12)
    for i, s_ in enumerate(many["spheres"]):
        if i % 3 == 0:
            s_.material.refraction, s_.material.refractive_index = 0.6, 1.4
    cases = [(refractive_scene(), 480, 270, 5), (cube_scene(glass=True), 320, 180, 4), (many, 480, 270, 4)]
    try:
        for scene, w, h, D in cases:
            out = []
            for on in (True, False):
                renderer.set_occluder_table(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                img, hit = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=5, return_hit_index=True)
                out.append((img.to(torch.int16), hit, renderer.ray_counts()))
            (a, ha, ca), (b, hb, cb) = out
            assert torch.equal(ha, hb) and ca == cb and ca["total"] > w * h
            d = (a - b).abs()
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    finally:
        renderer.set_occluder_table(True)
