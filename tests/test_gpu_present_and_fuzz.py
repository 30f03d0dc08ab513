"""(1) SURVEY.md 8(f) rank 4, presentation (raytracer.py:338-348): a frame rendered on the GPU goes through
raytracer_amd.present — the bytes glTexImage2D would be handed, an image file — and is still the reference's frame.
(2) Short forms of tools/fuzz_culls.py, fuzz_cubes.py and fuzz_refr_table.py, collected by pytest so that a cull or
table that changes a byte fails the GPU test run and not only a hand-run script.
(3) BASELINE config 4 on a pixel lattice of the 8K frame against the pinned oracle."""
import numpy as np
import pytest

from conftest import golden_scene_arrays, load_golden, scene_dict_from_arrays
from raytracer_amd.scenes import (AreaLight, Cube, Light, Material, Sphere, Vector3, default_scene, generate_scene,
                                  orbit_camera)

pytestmark = pytest.mark.gpu


def test_presentation_of_a_rendered_frame(renderer, tmp_path):
    import raytracer_amd
    from raytracer_amd import present
    from test_gpu_parity import check_rgb
    g = load_golden("c1_default_640x480_D3_modeK")
    renderer.upload_scene(default_scene())
    frame = renderer.render(640, 480, 1, max_depth=3, rng_mode="K", seed=0)          # a device tensor
    w, h, raw = present.texture_bytes(frame)
    assert (w, h) == (640, 480) and len(raw) == 640 * 480 * 3
    tex = np.frombuffer(raw, np.uint8).reshape(480, 640, 3)                           # row 0 = top, tightly packed RGB
    check_rgb(tex, g["rgb8"])
    path = present.save_image(frame, str(tmp_path / "frame.ppm"))
    data = open(path, "rb").read()
    assert data.startswith(b"P6\n640 480\n255\n") and data[len(b"P6\n640 480\n255\n"):] == raw
    # the drop-in render() (a host array, as the reference returns) presents the same bytes
    saved = raytracer_amd.scene
    try:
        raytracer_amd.set_scene(default_scene())
        host = raytracer_amd.render(640, 480, 1, max_depth=3, rng_mode="K", seed=0)
    finally:
        raytracer_amd.set_scene(saved)
    assert present.texture_bytes(host)[2] == raw


def test_config4_lattice_against_oracle(renderer, oracle):
    """BASELINE config 4 (7680x4320, 1024 spheres, depth 5): every 16th pixel of every 16th row of the full 8K frame —
    129 600 pixels, every region of the many-sphere visibility pass and of the list kernels — against the oracle."""
    from test_gpu_parity import check_hits, check_rgb
    g = load_golden("c4_1024sph_7680x4320_D5")
    m, a = g["meta"], golden_scene_arrays(g)
    w, h = m["width"], m["height"]
    scene = scene_dict_from_arrays(a)
    renderer.upload_scene(scene)
    img, hit = renderer.render(w, h, 1, max_depth=m["max_depth"], rng_mode=m["mode"], seed=m["seed"], return_hit_index=True)
    ys, xs = np.meshgrid(np.arange(8, h, 16), np.arange(8, w, 16), indexing="ij")
    px = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
    o = oracle.render_pixels(a["center"], a["radius"], a["color"], a["specular"], a["reflection"], a["lights"],
                             int(a["area_samples"]), w, h, pixels=px, samples=1, max_depth=m["max_depth"], mode=m["mode"],
                             seed=m["seed"])
    img, hit = img.cpu().numpy(), hit.cpu().numpy()
    assert len(px) == 270 * 480
    check_hits(hit[px[:, 1], px[:, 0]], o["hit"], o["margin"])
    check_rgb(img[px[:, 1], px[:, 0]], o["rgb8"])
    assert (o["hit"] >= 0).mean() > 0.2


def _random_lights(rng, sc, sph):
    def rand_pos():
        mode = rng.integers(0, 4)
        if mode == 0 or not sph:
            return Vector3(*(rng.uniform(-15, 15), rng.uniform(-10, 15), rng.uniform(-28, 12)))
        s = sph[int(rng.integers(0, len(sph)))]
        c = np.array([s.center.x, s.center.y, s.center.z]); d = rng.normal(size=3); d /= np.linalg.norm(d)
        return Vector3(*(c + d * s.radius * {1: 1.05, 2: 0.5, 3: 3.0}[int(mode)]))
    sc["global_light"] = Light(rand_pos(), Vector3(1, 1, 1), 0.9)
    ext = float(rng.choice([0.05, 0.5, 1.0, 3.0, 8.0]))
    u = rng.normal(size=3); v = rng.normal(size=3)
    sc["area_light"] = AreaLight(rand_pos(), Vector3(*(u / np.linalg.norm(u) * ext)),
                                 Vector3(*(v / np.linalg.norm(v) * ext * rng.uniform(0.2, 1))), Vector3(1, 1, 1), 0.9,
                                 samples=int(rng.choice([1, 4, 12, 20])))


@pytest.mark.parametrize("case", range(10))
def test_fuzz_culls_and_tables_change_no_byte(renderer, case):
    """tools/fuzz_culls.py in short: a random scene (3 .. 2040 spheres, lights next to / inside / far from spheres, a
    random camera) rendered by the default path, without the occluder table, without the direction table (plain scans of
    every sphere), and by the one-thread-per-pixel kernel (linear scans): identical frames and hit maps."""
    import torch
    rng = np.random.default_rng(1000 + case)
    n = int([3, 8, 20, 64, 150, 400, 1000, 2040, 64, 300][case])
    sc = generate_scene(n, int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    sph = sc["spheres"]
    if case % 4 == 1:      # a huge "floor" and a large far sphere: table rows of all ones
        sph.append(Sphere(Vector3(0, -1003.5, -10), 1000, Material(Vector3(0.8, 0.8, 0.8), specular=0.3, reflection=0.4)))
        sph.insert(0, Sphere(Vector3(25, 5, -45), 20, Material(Vector3(0.9, 0.5, 0.5), specular=0.3, reflection=0.6)))
    _random_lights(rng, sc, sph)
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h, D = int(rng.integers(100, 500)), int(rng.integers(60, 300)), int(rng.integers(0, 7))
    if n >= 1000:
        w, h = min(w, 240), min(h, 135)
    outs = []
    try:
        for table, kernel, refl in ((True, "stream", True), (False, "stream", True), (True, "stream", False), (False, "stream", False),
                                    (True, "pixel", True)):
            renderer.set_occluder_table(table); renderer.set_kernel(kernel); renderer.set_reflect_table(refl)
            renderer.upload_scene(sc)
            outs.append(renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True))
    finally:
        renderer.set_kernel("stream"); renderer.set_occluder_table(True); renderer.set_reflect_table(True)
    for i, o in enumerate(outs[1:]):
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]), (case, n, w, h, D, i + 1)


@pytest.mark.parametrize("case", range(10))
def test_fuzz_cubes_against_oracle(renderer, oracle, case):
    """tools/fuzz_cubes.py in short (that script compares digests of two BUILDS, with and without the candidate-box cull
    of lighting(); here the judge is the oracle): random scenes with boxes around lights, around the camera, on the lines
    from spheres to the lights on either side, opaque and of glass — frames within the bar, hit maps up to grazing ties."""
    from test_gpu_parity import check_hits, check_rgb
    rng = np.random.default_rng(2000 + case)
    n = int(rng.choice([0, 3, 8, 20]))
    sc = generate_scene(max(n, 1), int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    if n == 0:
        sc["spheres"] = []
    objs = list(sc["spheres"])
    glass = case % 2 == 1
    for s_ in objs:
        if glass and rng.random() < 0.3:
            s_.material.refraction = float(rng.uniform(0.2, 0.9)); s_.material.refractive_index = float(rng.choice([0.8, 1.33, 1.5]))
    lp = np.array([rng.uniform(-8, 8), rng.uniform(-4, 8), rng.uniform(-16, 8)])
    ap = np.array([rng.uniform(-8, 8), rng.uniform(-4, 8), rng.uniform(-16, 8)])
    for _ in range(int(rng.choice([1, 2, 5, 12]))):
        kind = int(rng.integers(0, 5))
        hh = rng.uniform(0.1, 1.5, size=3)
        balls = [o for o in objs if hasattr(o, "center")]
        if kind == 0 or not balls:
            c = np.array([rng.uniform(-8, 8), rng.uniform(-4, 8), rng.uniform(-16, 8)])
        elif kind == 1:
            c = (lp if rng.integers(0, 2) else ap) + rng.normal(size=3) * 0.7
        elif kind == 2:
            c = rng.normal(size=3) * 0.5; hh = rng.uniform(0.5, 3.0, size=3)
        else:
            s_ = balls[int(rng.integers(0, len(balls)))]
            sc_ = np.array([s_.center.x, s_.center.y, s_.center.z])
            c = sc_ + rng.uniform(-1.5, 2.0) * ((lp if rng.integers(0, 2) else ap) - sc_)
        m = Material(Vector3(*rng.uniform(0.1, 1.0, size=3)), specular=float(rng.uniform(0, 1)), reflection=float(rng.choice([0.0, 0.4])))
        if glass and rng.random() < 0.3:
            m.refraction = float(rng.uniform(0.2, 0.9)); m.refractive_index = float(rng.choice([1.0, 1.33, 1.5]))
        objs.insert(int(rng.integers(0, len(objs) + 1)), Cube(Vector3(*(c - hh)), Vector3(*(c + hh)), m))
    sc["spheres"] = objs
    sc["global_light"] = Light(Vector3(*lp), Vector3(1, 1, 1), 0.9)
    ext = float(rng.choice([0.05, 0.5, 1.0, 3.0]))
    u = rng.normal(size=3); v = rng.normal(size=3)
    sc["area_light"] = AreaLight(Vector3(*ap), Vector3(*(u / np.linalg.norm(u) * ext)), Vector3(*(v / np.linalg.norm(v) * ext * rng.uniform(0.2, 1))),
                                 Vector3(1, 1, 1), 0.9, samples=int(rng.choice([1, 4, 12])))
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h, D = int(rng.integers(100, 300)), int(rng.integers(60, 200)), int(rng.integers(0, 5))
    renderer.upload_scene(sc)
    img, hit = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
    o = oracle.render_scene(sc, w, h, samples=1, max_depth=D, mode="K", seed=case, camera=None if cam is None else cam.as_array())
    check_hits(hit.cpu().numpy(), o["hit"], o["margin"])
    check_rgb(img.cpu().numpy(), o["rgb8"])
    # the direction table of secondary rays with boxes as sources and targets (line rule, a box lists itself): the plain
    # scan of every object finds the same hits, so the frame is the same byte for byte (refractive sums are integers)
    import torch
    try:
        renderer.set_reflect_table(False); renderer.upload_scene(sc)
        img2, hit2 = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
    finally:
        renderer.set_reflect_table(True)
    assert torch.equal(hit, hit2) and torch.equal(img, img2), (case, n, len(objs), w, h, D)


@pytest.mark.parametrize("case", range(10))
def test_fuzz_refractive_tables_change_no_byte(renderer, case):
    """tools/fuzz_refr_table.py in short: random scenes with glass spheres (4 .. 250 spheres, a tenth to four fifths of
    glass), direction table on and off: the same tree — and, the sums being integers, the same bytes."""
    import torch
    rng = np.random.default_rng(3000 + case)
    n = int([4, 8, 20, 64, 150, 250, 8, 20, 64, 40][case])
    sc = generate_scene(n, int(rng.integers(1, 1000)), keep_default=bool(rng.integers(0, 2)) and n >= 3)
    frac = float(rng.choice([0.1, 0.35, 0.8]))
    for s_ in sc["spheres"]:
        if rng.random() < frac:
            s_.material.refraction = float(rng.uniform(0.2, 0.9))
            s_.material.refractive_index = float(rng.choice([0.8, 1.0, 1.33, 1.5, 2.2]))
    if not any(s_.material.refraction > 0 for s_ in sc["spheres"]):
        sc["spheres"][0].material.refraction = 0.5
    cam = None if rng.integers(0, 3) == 0 else orbit_camera(int(rng.integers(0, 256)), 256, radius=float(rng.uniform(3, 12)))
    w, h, D = int(rng.integers(100, 400)), int(rng.integers(60, 250)), int(rng.integers(1, 8))
    outs = []
    try:
        for on in (True, False):
            renderer.set_reflect_table(on); renderer.upload_scene(sc); renderer.reset_ray_counts()
            img, hit = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=case, camera=cam, return_hit_index=True)
            outs.append((img, hit, renderer.ray_counts()))
    finally:
        renderer.set_reflect_table(True)
    (a, ha, ca), (b, hb, cb) = outs
    assert ca == cb and torch.equal(ha, hb) and torch.equal(a, b), (case, n, w, h, D)
