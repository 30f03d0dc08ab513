"""Refractive scenes (raytracer.py:213-230, :249, :254, :258): the stream kernel sums the nodes of a
pixel's tree in its tile's slot — fixed point, in LDS — instead of float atomics into a slab in device memory.  Integer
addition is associative, so the bytes of a frame depend on the scene alone: not on the run, the work distribution, the
queue-spill setting, the tiling or the batch a frame is part of.  All through the C ABI."""
import numpy as np
import pytest

from raytracer_amd import tiling
from raytracer_amd.scenes import (Material, Sphere, Vector3, cube_scene, default_scene, generate_scene, orbit_camera,
                                  refractive_scene)

pytestmark = pytest.mark.gpu


def _glass40():
    many = generate_scene(40, 12)
    for i, s_ in enumerate(many["spheres"]):
        if i % 3 == 0:
            s_.material.refraction, s_.material.refractive_index = 0.6, 1.4
    return many


SCENES = {"glass_spheres": refractive_scene, "glass_cubes": lambda: cube_scene(glass=True), "glass40": _glass40}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_reruns_row_tiles_and_batches_are_byte_identical(renderer, name):
    import torch
    scene = SCENES[name]()
    renderer.upload_scene(scene)
    for (w, h, tr, G, D) in ((1920, 1080, 8, 2, 5), (640, 357, 8, 3, 5), (333, 187, 5, 4, 7)):
        renderer.reset_ray_counts()
        full, hit = renderer.render(w, h, 1, max_depth=D, seed=5, return_hit_index=True)
        ca = renderer.ray_counts()
        for _ in range(2):                                     # run to run: the same bytes
            renderer.reset_ray_counts()
            again, hit2 = renderer.render(w, h, 1, max_depth=D, seed=5, return_hit_index=True)
            assert torch.equal(full, again) and torch.equal(hit, hit2) and renderer.ray_counts() == ca
        assert ca["reflection"] > 0 and int(full.max()) > 50
        rows_max = tiling.max_local_rows(h, tr, G)             # row tiles of G ranks == the frame, byte for byte
        gathered = torch.zeros((G, rows_max * w * 3), dtype=torch.uint8, device="cuda:0")
        for r in range(G):
            loc = renderer.render_rows(w, h, 1, tile_rows=tr, tile_first=r, tile_stride=G, max_depth=D, seed=5)
            rows = torch.from_numpy(tiling.local_rows(h, tr, r, G)).cuda()
            assert torch.equal(loc, full[rows]), (name, w, h, r)
            gathered[r, :loc.numel()] = loc.reshape(-1)
        assert torch.equal(renderer.scatter_rows(gathered, G, w, h, tr), full)
    # a frame of a batch == the frame alone (other tiles share the wave's queue and slots)
    cams = [orbit_camera(3 * f, 64) for f in range(40)]
    batch = renderer.render_frames(cams, list(range(40)), 320, 180, 1, max_depth=5)
    for f in (0, 7, 39):
        assert torch.equal(batch[f], renderer.render(320, 180, 1, camera=cams[f], seed=f, max_depth=5))


def test_queue_spill_and_tables_change_no_byte(renderer):
    """Records set aside in device memory or not, direction table and occluder table on or off: the same tree, so the
    same sums — identical frames, hit maps and ray counts (deep trees: glass cubes at depth 5, 9 and 14)."""
    import torch
    cases = [(cube_scene(glass=True), 480, 270, 5), (cube_scene(glass=True), 240, 136, 9), (cube_scene(glass=True), 96, 54, 14),
             (refractive_scene(), 640, 360, 5), (_glass40(), 480, 270, 8)]
    try:
        for scene, w, h, D in cases:
            out = []
            for spill, table, occl in ((True, True, True), (False, True, True), (2, True, True), (3, False, True), (True, True, False)):
                renderer.set_queue_spill(spill); renderer.set_reflect_table(table); renderer.set_occluder_table(occl)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                img, hit = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=5, return_hit_index=True)
                out.append((img, hit, renderer.ray_counts()))
            a, ha, ca = out[0]
            assert ca["total"] > w * h
            for b, hb, cb in out[1:]:
                assert torch.equal(a, b) and torch.equal(ha, hb) and ca == cb, (w, h, D)
    finally:
        renderer.set_queue_spill(True); renderer.set_reflect_table(True); renderer.set_occluder_table(True)


def test_slot_sums_agree_with_float_sums(renderer, oracle):
    """RT_OPT_TILE_FLUSH off keeps the earlier form (float atomics into a slab, resolve pass): the same trees, sums that
    differ in their last bits only — at most 1/255 on a few pixels — and both within the bar of the oracle."""
    import torch
    from test_gpu_parity import check_rgb
    for scene, w, h, D in ((refractive_scene(), 640, 360, 5), (cube_scene(glass=True), 333, 187, 5)):
        out = []
        try:
            for on in (True, False):
                renderer.set_tile_flush(on)
                renderer.upload_scene(scene)
                renderer.reset_ray_counts()
                img = renderer.render(w, h, 1, max_depth=D, rng_mode="K", seed=3)
                out.append((img.to(torch.int16), renderer.ray_counts()))
        finally:
            renderer.set_tile_flush(True)
        (a, ca), (b, cb) = out
        assert ca == cb
        d = (a - b).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
        o = oracle.render_scene(scene, w, h, samples=1, max_depth=D, mode="K", seed=3)
        check_rgb(a.cpu().numpy(), o["rgb8"])


def test_bright_lights_saturate_and_negative_colours_fall_back(renderer, oracle):
    """The fixed-point sums saturate at 256 (far beyond the clamp at 1: the pixel is white, as in the reference), and a
    scene whose colours can make a node's share negative is summed in float as before (the library decides at
    upload): both against the oracle."""
    from test_gpu_parity import check_rgb
    bright = refractive_scene()
    bright["global_light"].intensity = 400.0
    bright["area_light"].intensity = 900.0
    odd = refractive_scene()
    odd["spheres"][0].material.color = Vector3(-0.4, 0.9, 0.5)
    for scene in (bright, odd):
        renderer.upload_scene(scene)
        img = renderer.render(320, 180, 1, max_depth=5, rng_mode="K", seed=2).cpu().numpy()
        o = oracle.render_scene(scene, 320, 180, samples=1, max_depth=5, mode="K", seed=2)
        check_rgb(img, o["rgb8"])
        assert img.max() == 255


@pytest.mark.parametrize("name", sorted(SCENES))
def test_samples_in_one_launch_on_refractive_scenes(renderer, oracle, name):
    """2 .. 64 samples per pixel of a refractive scene are shaded by ONE stream-kernel launch, every sample's tree summed
    into the tile's slot (render_pixel :292-297): byte-identical run to run and across row tiles, within the bar of the
    oracle, and — mode C, a power of two of samples: every sample is the same tree — identical to the one-sample frame."""
    import torch
    from test_gpu_parity import check_rgb
    scene = SCENES[name]()
    renderer.upload_scene(scene)
    w, h, D = 333, 187, 5
    for spp in (2, 3, 8):
        a = renderer.render(w, h, spp, max_depth=D, rng_mode="K", seed=6)
        assert torch.equal(a, renderer.render(w, h, spp, max_depth=D, rng_mode="K", seed=6))
        rows = renderer.render_rows(w, h, spp, tile_rows=8, tile_first=1, tile_stride=3, max_depth=D, rng_mode="K", seed=6)
        assert torch.equal(rows, a[torch.from_numpy(tiling.local_rows(h, 8, 1, 3)).cuda()])
        if spp <= 3:
            o = oracle.render_scene(scene, w, h, samples=spp, max_depth=D, mode="K", seed=6)
            check_rgb(a.cpu().numpy(), o["rgb8"])
    one = renderer.render(w, h, 1, max_depth=D, rng_mode="C")
    d = (one.to(torch.int16) - renderer.render(w, h, 4, max_depth=D, rng_mode="C").to(torch.int16)).abs()
    assert int(d.max()) == 0          # (4 x the same integers, scaled by 2^-2: exact)
    # the float-slab path (RT_OPT_TILE_FLUSH off) agrees up to the last bit of the sums
    try:
        renderer.set_tile_flush(False)
        b = renderer.render(w, h, 3, max_depth=D, rng_mode="K", seed=6)
    finally:
        renderer.set_tile_flush(True)
    d = (renderer.render(w, h, 3, max_depth=D, rng_mode="K", seed=6).to(torch.int16) - b.to(torch.int16)).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01


def test_many_frames_at_64_samples_stay_within_the_record_budget(renderer):
    """A batch whose tile records (144 B per tile, frame and sample) would exceed the launch's budget is split into several
    launches instead of one allocation of hundreds of GB: 100 frames of 1080p at 64 samples per pixel; frames of both
    launches equal their single renders."""
    import torch
    renderer.upload_scene(default_scene())
    n = 100
    cams = [orbit_camera(f, 128) for f in range(n)]
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    batch = renderer.render_frames(cams, list(range(n)), 1920, 1080, 64, max_depth=3, rng_mode="K")
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < (40 << 30)
    for f in (0, 80, 81, 99):
        assert torch.equal(batch[f], renderer.render(1920, 1080, 64, camera=cams[f], seed=f, max_depth=3, rng_mode="K"))
