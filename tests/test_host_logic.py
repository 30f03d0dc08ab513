"""Host-side logic that needs no GPU: scene packing, generators, tiling, ABI surface."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from raytracer_amd import _lib, tiling
from raytracer_amd.scenes import (Camera, RtCamera, RtLights, default_scene, generate_scene,
                                  orbit_camera, pack_scene, scene_from_arrays)


def test_pack_default_scene_layout():
    p = pack_scene(default_scene())
    assert p.geom4.shape == (3, 4) and p.mat4.shape == (3, 4) and p.spec.shape == (3,)
    assert p.geom4.dtype == np.float32
    assert np.allclose(p.geom4[1], [-2.5, 0, -7, 2.25])           # cx cy cz r^2
    assert np.allclose(p.mat4[2], [0.35, 1, 1, 0.7])              # R G B reflection
    assert p.area_samples == 12 and np.allclose(p.lights[:7], [0, 10, 10, 1, 1, 1, 0.9])
    L = p.ct_lights()
    assert L.area_samples == 12 and abs(L.area_intensity - 0.9) < 1e-6 and L.area_pos[0] == 5.0


def test_scene_roundtrip_is_float32_exact():
    p = pack_scene(generate_scene(16, 5))
    s2 = scene_from_arrays(p.geom4[:, :3], p.radius, p.mat4[:, :3], p.spec, p.mat4[:, 3], p.lights, 12)
    p2 = pack_scene(s2)
    for a, b in ((p.geom4, p2.geom4), (p.mat4, p2.mat4), (p.spec, p2.spec), (p.lights, p2.lights)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("n,seed,keep", [(8, 1, True), (64, 2, False), (1024, 3, False)])
def test_generated_scenes(n, seed, keep):
    a = pack_scene(generate_scene(n, seed, keep_default=keep))
    b = pack_scene(generate_scene(n, seed, keep_default=keep))
    assert a.n_spheres == n and np.array_equal(a.geom4, b.geom4)            # deterministic
    c, r = a.geom4[:, :3].astype(np.float64), a.radius.astype(np.float64)
    d = np.linalg.norm(c[:, None] - c[None], axis=-1) + np.eye(n) * 1e9
    assert (d > r[:, None] + r[None]).all()                                   # no overlaps
    assert (np.linalg.norm(c, axis=1) > r).all()                              # camera outside
    assert (a.mat4[:, 3] >= 0.2).all() and (a.mat4[:, 3] <= 0.8).all()
    if keep:
        assert np.array_equal(a.geom4[:3], pack_scene(default_scene()).geom4)


def test_limits_are_enforced():
    s = default_scene()
    s["area_light"].samples = 0
    with pytest.raises(ValueError):
        pack_scene(s)
    s["area_light"].samples = 65
    with pytest.raises(ValueError):
        pack_scene(s)


def test_orbit_camera_is_orthonormal_and_looks_at_centre():
    for f in (0, 1, 64, 200, 255):
        a = orbit_camera(f).as_array().astype(np.float64)
        o, r, u, fw = a
        assert abs(np.linalg.norm(r) - 1) < 1e-6 and abs(np.linalg.norm(fw) - 1) < 1e-6
        assert abs(r @ u) < 1e-6 and abs(r @ fw) < 1e-6 and abs(u @ fw) < 1e-6
        to_c = np.array([0, 0, -6.0]) - o
        assert np.allclose(to_c / np.linalg.norm(to_c), fw, atol=1e-6)
    d = Camera().as_array()
    assert np.array_equal(d, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, -1]])


@pytest.mark.parametrize("height,tile_rows,world", [(1080, 8, 1), (1080, 8, 2), (1080, 8, 8),
                                                     (4320, 16, 4), (37, 8, 3), (5, 8, 4), (1, 1, 2)])
def test_row_tiles_partition_the_image(height, tile_rows, world):
    seen = np.concatenate([tiling.local_rows(height, tile_rows, r, world) for r in range(world)])
    assert np.array_equal(np.sort(seen), np.arange(height))                   # each row exactly once
    counts = [tiling.local_row_count(height, tile_rows, r, world) for r in range(world)]
    assert max(counts) == tiling.max_local_rows(height, tile_rows, world)
    assert max(counts) - min(counts) <= tile_rows                             # balanced


def test_ctypes_structs_match_header():
    assert ctypes.sizeof(RtCamera) == 48
    assert ctypes.sizeof(RtLights) == 84
    hdr = open(os.path.join(ROOT, "include", "rt355.h")).read()
    assert "float origin[3];" in hdr and "int32_t area_samples;" in hdr


def test_shared_library_exports_every_declared_symbol():
    """include/rt355.h <-> librt355.so <-> the ctypes table agree (no compute call: no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "rt355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.library_path()], text=True)
    exported = set(re.findall(r" T (rt_[a-z_]+)", out))
    assert declared <= exported
    assert L.rt_version().startswith(b"rt355")


def test_no_gpu_means_loud_failure():
    """Without a GPU the product path must fail, never fall back to a CPU renderer."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.rt_create(0, ctypes.byref(h))
    assert rc == -2 and b"no CPU path" in L.rt_last_error(None)
    import raytracer_amd
    with pytest.raises(RuntimeError):
        raytracer_amd.render(4, 4)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").lower() or f in ("__init__.py",) \
                    or "import oracle" not in src and "from oracle" not in src, f
                assert "import oracle" not in src and "from oracle" not in src, f


def test_presentation_helpers(tmp_path):
    """the frame is what glTexImage2D / an image file takes."""
    from raytracer_amd import present
    img = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    w, h, raw = present.texture_bytes(img[:, ::1])
    assert (w, h) == (7, 5) and raw == img.tobytes()
    p = present.save_image(img, str(tmp_path / "f.ppm"))
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n7 5\n255\n") and data.endswith(img.tobytes())
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        q = present.save_image(img, str(tmp_path / "f.png"))
        assert np.array_equal(np.asarray(Image.open(q)), img)
    with pytest.raises(ValueError):
        present.as_host_rgb(img.astype(np.float32))


def test_occluder_table_is_a_superset_of_brute_force():
    """rt_occluder_table (host-only, FP64): every sphere that a shadow ray leaving sphere k towards the
    point light or any point of the area light can hit must have its bit set in row k.  Brute force:
    random points on each sphere, offset 0.001 along the normal (raytracer.py:184, :199), rays to the
    point light and to the corners / random points of the area light, intersect_sphere (:93-104)."""
    import ctypes
    from raytracer_amd import _lib
    from raytracer_amd.scenes import generate_scene, pack_scene, default_scene
    L = _lib.load()
    rng = np.random.default_rng(3)
    for scene in (default_scene(), generate_scene(8, 1, keep_default=True), generate_scene(64, 2), generate_scene(150, 7)):
        p = pack_scene(scene)
        n, words = p.n_spheres, (p.n_spheres + 31) // 32
        out = np.zeros((n, words), np.uint32)
        lights = p.ct_lights()
        g = np.ascontiguousarray(p.geom4, np.float32)
        assert L.rt_occluder_table(g.ctypes.data, n, ctypes.byref(lights), out.ctypes.data) == 0
        bits = ((out[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :n].astype(bool)
        C = p.geom4[:, :3].astype(np.float64); r = np.sqrt(p.geom4[:, 3].astype(np.float64))
        lv = p.lights.astype(np.float64)
        ppos, apos, au, av = lv[0:3], lv[7:10], lv[10:13], lv[13:16]
        targets = [ppos] + [apos + au * a + av * b for a in (-0.5, 0.5) for b in (-0.5, 0.5)] + \
                  [apos + au * (rng.random() - 0.5) + av * (rng.random() - 0.5) for _ in range(6)]
        hit_any = np.zeros((n, n), bool)
        for k in range(n):
            d = rng.normal(size=(400, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
            S = C[k] + d * (r[k] + 0.001)
            for T in targets:
                Ld = T - S; Ld /= np.linalg.norm(Ld, axis=1, keepdims=True)
                oc = S[:, None, :] - C[None, :, :]                     # [pts, spheres, 3]
                b = 2 * np.einsum("psk,pk->ps", oc, Ld)
                c = np.einsum("psk,psk->ps", oc, oc) - r[None, :] ** 2
                disc = b * b - 4 * c
                t = (-b - np.sqrt(np.maximum(disc, 0))) / 2
                hit_any[k] |= ((disc >= 0) & (t > 0)).any(axis=0)
        assert not (hit_any & ~bits).any(), "a sphere that is hit is missing from the table"
        if n >= 64:      # and the table is selective: a few candidates per object, not all of them
            assert bits.sum(1).mean() < 0.25 * n


def _bucket_f32(Df, B):
    """direction_bucket() of rt_core.hpp, in FP32 as the kernel computes it."""
    m = len(Df)
    a = np.abs(Df); mx = (a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]); my = ~mx & (a[:, 1] >= a[:, 2])
    mi = np.where(mx, 0, np.where(my, 1, 2)); idx = np.arange(m)
    major = Df[idx, mi]; inv = np.float32(1.0) / np.abs(major)
    hB = np.float32(0.5 * B)
    ix = np.clip((Df[idx, (mi + 1) % 3] * inv * hB + hB).astype(np.int32), 0, B - 1)
    iy = np.clip((Df[idx, (mi + 2) % 3] * inv * hB + hB).astype(np.int32), 0, B - 1)
    return ((2 * mi + (major < 0)) * B + iy) * B + ix


def test_occluder_buckets_are_a_superset_of_brute_force():
    """rt_occluder_table_buckets (host-only, FP64): rows per (sphere k, bucket of the surface normal).  Brute force:
    random points on each sphere (normal N, shadow origin 0.001 along it, raytracer.py:184, :199), rays to the point
    light and to corners / random points of the area light, intersect_sphere (:93-104) against EVERY sphere, k itself
    included: whatever is hit must be listed in the row of (k, bucket of N) — the bucket computed in FP32 as the kernel
    does.  Also: the rows are far emptier than the per-object rows, and points that face both lights list nothing."""
    import ctypes
    from raytracer_amd import _lib
    from raytracer_amd.scenes import AreaLight, Light, Vector3, generate_scene, pack_scene, default_scene
    L = _lib.load()
    rng = np.random.default_rng(11)
    near = generate_scene(20, 5)        # lights next to and inside spheres: the "rays go anywhere" rows
    s0 = near["spheres"][0]
    near["global_light"] = Light(Vector3(s0.center.x + s0.radius * 1.02, s0.center.y, s0.center.z), Vector3(1, 1, 1), 0.9)
    near["area_light"] = AreaLight(Vector3(near["spheres"][3].center.x, near["spheres"][3].center.y, near["spheres"][3].center.z),
                                   Vector3(3, 0, 0), Vector3(0, 2, 1), Vector3(1, 1, 1), 0.9, samples=12)
    for scene, B in ((default_scene(), 16), (generate_scene(8, 1, keep_default=True), 16), (generate_scene(64, 2), 16),
                     (generate_scene(64, 2), 4), (generate_scene(150, 7), 8), (near, 8)):
        p = pack_scene(scene)
        n, words, nb = p.n_spheres, (p.n_spheres + 31) // 32, 6 * B * B
        out = np.zeros((n, nb, words), np.uint32)
        lights = p.ct_lights()
        g = np.ascontiguousarray(p.geom4, np.float32)
        assert L.rt_occluder_table_buckets(g.ctypes.data, n, ctypes.byref(lights), B, out.ctypes.data) == 0
        bits = ((out[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, nb, -1)[:, :, :n].astype(bool)
        obj = np.zeros((n, words), np.uint32)
        assert L.rt_occluder_table(g.ctypes.data, n, ctypes.byref(lights), obj.ctypes.data) == 0
        obits = ((obj[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :n].astype(bool)
        assert not (bits & ~obits[:, None, :]).any()                   # a refinement of the per-object rows
        C = p.geom4[:, :3].astype(np.float64); r = np.sqrt(p.geom4[:, 3].astype(np.float64))
        lv = p.lights.astype(np.float64)
        ppos, apos, au, av = lv[0:3], lv[7:10], lv[10:13], lv[13:16]
        targets = [ppos] + [apos + au * a + av * b for a in (-0.5, 0.5) for b in (-0.5, 0.5)] + \
                  [apos + au * (rng.random() - 0.5) + av * (rng.random() - 0.5) for _ in range(6)]
        missing = 0
        for k in range(n):
            m = 1500
            nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            S = C[k] + nrm * (r[k] + 0.001)
            bucket = _bucket_f32(nrm.astype(np.float32), B)
            for T in targets:
                Ld = T - S; Ld /= np.linalg.norm(Ld, axis=1, keepdims=True)
                oc = S[:, None, :] - C[None, :, :]
                b = 2 * np.einsum("psk,pk->ps", oc, Ld)
                c = np.einsum("psk,psk->ps", oc, oc) - r[None, :] ** 2
                disc = b * b - 4 * c
                t = (-b - np.sqrt(np.maximum(disc, 0))) / 2
                hit = (disc >= 0) & (t > 0)
                missing += int((hit & ~bits[k, bucket]).sum())
        assert missing == 0, "a sphere that a shadow ray hits is missing from the row of its (object, normal bucket)"
        if scene is not near and B >= 8:
            print(n, B, bits.sum(2).mean(), obits.sum(1).mean(), (bits.sum(2) == 0).mean())
            assert bits.sum(2).mean() < 0.6 * obits.sum(1).mean()      # emptier than the per-object rows
            assert (bits.sum(2) == 0).mean() > 0.15                    # and a good share of the rows lists nothing at all


def test_reflect_table_is_a_superset_of_brute_force():
    """rt_reflect_table (host-only, FP64): bit j of entry [k][bucket] must be set for every ray that leaves sphere k
    (random surface points, offset 0.001 along the normal, raytracer.py:247; random outward directions — every
    reflected direction is one) with a direction in the bucket and hits sphere j (intersect_sphere :93-104).  The
    bucket of a direction is computed here in FP32, as the kernel does it."""
    import ctypes
    from raytracer_amd import _lib
    from raytracer_amd.scenes import generate_scene, pack_scene
    L = _lib.load()
    rng = np.random.default_rng(5)
    for scene, B in ((generate_scene(16, 9), 16), (generate_scene(64, 2), 16), (generate_scene(64, 2), 4), (generate_scene(150, 7), 8)):
        p = pack_scene(scene)
        n, pairs, nb = p.n_spheres, (p.n_spheres + 63) // 64, 6 * B * B
        out = np.zeros((n, nb, pairs * 2), np.uint32)
        g = np.ascontiguousarray(p.geom4, np.float32)
        assert L.rt_reflect_table(g.ctypes.data, n, B, out.ctypes.data) == 0
        bits = ((out[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, nb, -1)[:, :, :n].astype(bool)
        assert not bits[np.arange(n), :, np.arange(n)].any()           # an object never lists itself
        C = p.geom4[:, :3].astype(np.float64); r = np.sqrt(p.geom4[:, 3].astype(np.float64))
        missing = 0
        for k in range(n):
            m = 3000
            nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            O = C[k] + nrm * (r[k] + 0.001)
            D = rng.normal(size=(m, 3)); D /= np.linalg.norm(D, axis=1, keepdims=True)
            D = np.where((np.sum(D * nrm, 1) < 0)[:, None], -D, D)    # outward, like a reflected ray
            # aim half of the rays at other spheres (random points in them), so that hits are not rare
            tgt = rng.integers(0, n, m); aim = C[tgt] + rng.normal(size=(m, 3)) * (r[tgt][:, None] * 0.6)
            Da = aim - O; Da /= np.linalg.norm(Da, axis=1, keepdims=True)
            use = (rng.random(m) < 0.5) & (tgt != k) & (np.sum(Da * nrm, 1) > 0)
            D = np.where(use[:, None], Da, D)
            Df = D.astype(np.float32)
            a = np.abs(Df); mx = (a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]); my = ~mx & (a[:, 1] >= a[:, 2])
            mi = np.where(mx, 0, np.where(my, 1, 2)); idx = np.arange(m)
            major = Df[idx, mi]; inv = np.float32(1.0) / np.abs(major)
            hB = np.float32(0.5 * B)
            ix = np.clip((Df[idx, (mi + 1) % 3] * inv * hB + hB).astype(np.int32), 0, B - 1)
            iy = np.clip((Df[idx, (mi + 2) % 3] * inv * hB + hB).astype(np.int32), 0, B - 1)
            bucket = ((2 * mi + (major < 0)) * B + iy) * B + ix
            oc = O[:, None, :] - C[None, :, :]
            b = 2 * np.einsum("psk,pk->ps", oc, D)
            c = np.einsum("psk,psk->ps", oc, oc) - r[None, :] ** 2
            disc = b * b - 4 * c
            tt = (-b - np.sqrt(np.maximum(disc, 0))) / 2
            hit = (disc >= 0) & (tt > 0)
            hit[:, k] = False
            missing += int((hit & ~bits[k, bucket]).sum())
        assert missing == 0, "a sphere that a ray hits is missing from its bucket"
        if B >= 8:       # and the table is selective: a few spheres per bucket, not all of them
            assert bits.sum(2).mean() < 0.1 * n


def test_pack_scene_enforces_the_cube_limit_before_the_device_is_touched():
    """rt_set_cubes accepts n_spheres + 2 n_cubes <= RT_MAX_SPHERES; pack_scene raises for anything beyond, so
    Renderer.upload_scene cannot fail between rt_upload_scene and rt_set_cubes on a scene that is merely too large."""
    from raytracer_amd.scenes import MAX_SPHERES, Cube, Material, Sphere, Vector3, default_scene, pack_scene
    m = Material(Vector3(1, 1, 1))
    sc = default_scene()
    sc["spheres"] = [Sphere(Vector3(i, 0, -5), 0.1, m) for i in range(MAX_SPHERES - 800)] + \
                    [Cube(Vector3(i, 1, -5), Vector3(i + 0.5, 1.5, -4.5), m) for i in range(400)]
    pack_scene(sc)                                   # 1248 + 800 = 2048: fits
    sc["spheres"].append(Cube(Vector3(0, 3, -5), Vector3(1, 4, -4), m))
    with pytest.raises(ValueError, match="2 x cubes"):
        pack_scene(sc)
