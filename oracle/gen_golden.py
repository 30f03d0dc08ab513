"""TEST INFRASTRUCTURE ONLY: generate tests/golden/*.npz by running the REFERENCE ITSELF
(/root/reference/raytracer.py through oracle/ref_harness.py) in the build container.

    python -m oracle.gen_golden [--only NAME ...] [--jobs 8]

The fixtures are data only: scene arrays, pixel coordinates, RNG mode/seed, and the reference's
outputs (FP64 colour before clamping, the quantised uint8 triple, primary hit index, per-pixel
trace/shade level counts).  No reference source text is stored.  The C++ oracle
(oracle/rt_oracle.cpp) is used here only to CHOOSE which pixels of the huge frames to sample
(half of them on spheres / silhouettes, which uniform sampling would mostly miss); every stored
value comes from the reference.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cpu_oracle, ref_harness  # noqa: E402
from raytracer_amd.scenes import (cube_scene, refractive_scene, Camera, default_scene, generate_scene, orbit_camera,  # noqa: E402
                                 pack_scene)

GOLD = os.path.join(ROOT, "tests", "golden")


def _worker(job):
    scene_kw, width, height, pixels, samples, D, mode, seed, cam = job
    scene = None if scene_kw is None else _scene_from_kw(scene_kw)
    camera = None if cam is None else Camera(*[tuple(r) for r in cam])
    return ref_harness.render_pixels(scene, width, height, pixels, samples, D, mode, seed, camera)


def _scene_from_kw(kw):
    from raytracer_amd.scenes import scene_from_arrays
    return scene_from_arrays(kw["center"], kw["radius"], kw["color"], kw["specular"],
                             kw["reflection"], kw["lights"], kw["area_samples"],
                             refraction=kw.get("refraction"), refractive_index=kw.get("refractive_index"),
                             kind=kw.get("kind"), box_max=kw.get("box_max"))


def scene_kw(scene):
    p = pack_scene(scene)
    if p.n_cubes:      # the object list in the reference's order: center = min_point for cubes
        o = p.objects_in_list_order()
        return dict(center=o["p0"].copy(), radius=o["radius"].copy(), color=o["mat4"][:, :3].copy(),
                    specular=o["spec"].copy(), reflection=o["mat4"][:, 3].copy(), lights=p.lights.copy(),
                    area_samples=p.area_samples, refraction=o["refr"][:, 0].copy(),
                    refractive_index=o["refr"][:, 1].copy(), kind=o["kind"].copy(), box_max=o["p1"].copy())
    return dict(center=p.geom4[:, :3].copy(), radius=p.radius.copy(), color=p.mat4[:, :3].copy(),
                specular=p.spec.copy(), reflection=p.mat4[:, 3].copy(), lights=p.lights.copy(),
                area_samples=p.area_samples,
                **(dict(refraction=p.refr[:, 0].copy(), refractive_index=p.refr[:, 1].copy())
                   if p.has_refraction else {}))


def run_reference(pool, skw, width, height, pixels, samples, D, mode, seed, cam=None, chunks=64):
    pixels = np.asarray(pixels, np.int32).reshape(-1, 2)
    parts = np.array_split(pixels, max(1, min(chunks, len(pixels))))
    cam_l = None if cam is None else cam.as_array().tolist()
    jobs = [(skw, width, height, p, samples, D, mode, seed, cam_l) for p in parts if len(p)]
    res = pool.map(_worker, jobs)
    return {k: np.concatenate([r[k] for r in res]) for k in res[0]}


def choose_pixels(skw, width, height, D, seed, n, rng, cam=None):
    """n pixels: 1/4 uniform, 1/2 on spheres, 1/4 on hit-index boundaries (chosen with the C++
    oracle's primary-hit map at mode C; selection only)."""
    full_w, full_h, sub = width, height, 1
    while width * height > 2_500_000:          # select on a coarser grid for the 4K/8K frames
        width, height, sub = width // 2, height // 2, sub * 2
    o = cpu_oracle.render_pixels(skw["center"], skw["radius"], skw["color"], skw["specular"],
                                 skw["reflection"], skw["lights"], skw["area_samples"], width,
                                 height, max_depth=0, mode="C", seed=seed,
                                 camera=None if cam is None else cam.as_array())
    hit = o["hit"].reshape(height, width)
    edge = np.zeros_like(hit, bool)
    edge[:, 1:] |= hit[:, 1:] != hit[:, :-1]
    edge[1:, :] |= hit[1:, :] != hit[:-1, :]
    out = [np.stack([rng.integers(0, width, n // 4), rng.integers(0, height, n // 4)], 1)]
    for mask, k in ((hit >= 0, n // 2), (edge, n - n // 4 - n // 2)):
        ys, xs = np.nonzero(mask)
        if len(ys) == 0:
            ys, xs = np.nonzero(np.ones_like(mask))
        sel = rng.choice(len(ys), size=min(k, len(ys)), replace=False)
        out.append(np.stack([xs[sel], ys[sel]], 1))
    px = np.concatenate(out).astype(np.int64)
    if sub > 1:
        px = px * sub + rng.integers(0, sub, px.shape)
        px[:, 0] = np.minimum(px[:, 0], full_w - 1)
        px[:, 1] = np.minimum(px[:, 1], full_h - 1)
    return np.unique(px.astype(np.int32), axis=0)


def save(name, meta, skw, res, pixels=None, extra=None):
    d = dict(meta=np.array(json.dumps(meta)))
    if skw is not None:
        d.update({"scene_" + k: np.asarray(v) for k, v in skw.items()})
    if pixels is not None:
        d["pixels"] = np.asarray(pixels, np.int32)
    d.update(res)
    if extra:
        d.update(extra)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)", flush=True)


def full_frame_pixels(w, h):
    ys, xs = np.mgrid[0:h, 0:w]
    return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)


def compact_full(res, w, h):
    """Full frames: keep the uint8 image, int8 hit map, uint8 level counts and float32 colour."""
    return dict(rgb8=res["rgb8"].reshape(h, w, 3), hit=res["hit"].reshape(h, w).astype(np.int16),
                trace_levels=res["trace_levels"].reshape(h, w).astype(np.uint8),
                hit_levels=res["hit_levels"].reshape(h, w).astype(np.uint8),
                color=res["color"].reshape(h, w, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--jobs", type=int, default=8)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    cpu_oracle.build()
    want = lambda n: args.only is None or n in args.only
    pool = mp.Pool(args.jobs)
    rng = np.random.default_rng(20241016)
    t00 = time.time()

    # --- harness self-check: the reference's native render() under random.seed(0) -------------
    if want("native_sha"):
        t0 = time.time()
        img = ref_harness.reference_render_native(640, 480, 1, seed=0)
        sha = hashlib.sha1(img.tobytes()).hexdigest()
        info = dict(sha1=sha, mean_rgb=img.reshape(-1, 3).mean(0).tolist(),
                    hit_fraction=float((img.reshape(-1, 3).max(1) > 0).mean()),
                    note="reference render(640,480,1) with random.seed(0), untouched MT stream; "
                         "SURVEY.md §8c recorded c4847bd063139e0f3b964c45f33b6a6d88c97a13")
        with open(os.path.join(GOLD, "native_mt_seed0_640x480.json"), "w") as f:
            json.dump(info, f, indent=1)
        print("native_sha", sha, f"{time.time() - t0:.1f}s", flush=True)

    # --- C1: the reference's OWN (unrounded) module-global scene, 640x480, D=3 -----------------
    for mode in ("C", "K"):
        name = f"c1_default_640x480_D3_mode{mode}"
        if want(name):
            t0 = time.time()
            px = full_frame_pixels(640, 480)
            res = run_reference(pool, None, 640, 480, px, 1, 3, mode, 0)
            meta = dict(width=640, height=480, samples=1, max_depth=3, mode=mode, seed=0,
                        scene="reference module-global scene (raytracer.py:78-87), unrounded")
            save(name, meta, None, compact_full(res, 640, 480))
            print(name, f"{time.time() - t0:.1f}s", flush=True)

    # --- pin: float32-rounded default scene, small frames, D sweep, spp 2 (bit-exact pinning) --
    for D in (0, 1, 3, 5, 8):
        name = f"pin_default_160x120_D{D}_spp2_modeK"
        if want(name):
            skw = scene_kw(default_scene())
            px = full_frame_pixels(160, 120)
            res = run_reference(pool, skw, 160, 120, px, 2, D, "K", 7)
            meta = dict(width=160, height=120, samples=2, max_depth=D, mode="K", seed=7,
                        scene="default scene rounded to float32")
            save(name, meta, skw, compact_full(res, 160, 120))

    # --- refraction: glass added to the default scene, full small frames ----
    for mode, D, spp in (("K", 5, 2), ("C", 3, 1)):
        name = f"refr_glass_200x150_D{D}_spp{spp}_mode{mode}"
        if want(name):
            t0 = time.time()
            skw = scene_kw(refractive_scene())
            px = full_frame_pixels(200, 150)
            res = run_reference(pool, skw, 200, 150, px, spp, D, mode, 3)
            meta = dict(width=200, height=150, samples=spp, max_depth=D, mode=mode, seed=3,
                        scene="raytracer_amd.scenes.refractive_scene() rounded to float32")
            save(name, meta, skw, compact_full(res, 200, 150))
            print(name, f"{time.time() - t0:.1f}s", flush=True)

    # --- cubes: Cube objects in the reference's object list ---------------------
    for glass, mode, D, spp in ((False, "K", 5, 2), (True, "C", 3, 1), (True, "K", 5, 1)):
        name = f"cube_{'glass' if glass else 'opaque'}_200x150_D{D}_spp{spp}_mode{mode}"
        if want(name):
            t0 = time.time()
            skw = scene_kw(cube_scene(glass))
            px = full_frame_pixels(200, 150)
            res = run_reference(pool, skw, 200, 150, px, spp, D, mode, 5)
            meta = dict(width=200, height=150, samples=spp, max_depth=D, mode=mode, seed=5,
                        scene=f"raytracer_amd.scenes.cube_scene(glass={glass}) rounded to float32")
            save(name, meta, skw, compact_full(res, 200, 150))
            print(name, f"{time.time() - t0:.1f}s", flush=True)

    # --- C2..C4: sparse pixel subsets of the big frames ---------------------------------------
    cfgs = [
        ("c2_8sph_1920x1080_D5", dict(n=8, sseed=1, keep_default=True), 1920, 1080, 5, 4096, 1),
        ("c3_64sph_3840x2160_D8", dict(n=64, sseed=2, keep_default=False), 3840, 2160, 8, 4096, 1),
        ("c4_1024sph_7680x4320_D5", dict(n=1024, sseed=3, keep_default=False), 7680, 4320, 5, 1024, 1),
        ("t64_64sph_1920x1080_D5", dict(n=64, sseed=2, keep_default=False), 1920, 1080, 5, 2048, 1),
        ("main_default_2560x1440_D5_spp2", None, 2560, 1440, 5, 2048, 2),
    ]
    for name, gen, w, h, D, npx, spp in cfgs:
        if not want(name):
            continue
        t0 = time.time()
        scene = default_scene() if gen is None else generate_scene(gen["n"], gen["sseed"], w, h,
                                                                   gen["keep_default"])
        skw = scene_kw(scene)
        px = choose_pixels(skw, w, h, D, 0, npx, rng)
        res = run_reference(pool, skw, w, h, px, spp, D, "K", 0)
        meta = dict(width=w, height=h, samples=spp, max_depth=D, mode="K", seed=0,
                    scene=f"generate_scene({gen})" if gen else "default scene rounded to float32")
        save(name, meta, skw, res, px)
        print(name, len(px), "pixels", f"{time.time() - t0:.1f}s",
              "hit frac", float((res['hit'] >= 0).mean()), flush=True)

    # --- C5: orbit-camera frames over the C2 scene; mode K seed = frame -----------------------
    name = "c5_orbit_1920x1080_D5"
    if want(name):
        t0 = time.time()
        scene = generate_scene(8, 1, 1920, 1080, True)
        skw = scene_kw(scene)
        frames, allpx, allres, cams = [0, 37, 128, 200], [], [], []
        for f in frames:
            cam = orbit_camera(f, 256)
            px = choose_pixels(skw, 1920, 1080, 5, f, 768, rng, cam)
            res = run_reference(pool, skw, 1920, 1080, px, 1, 5, "K", f, cam)
            allpx.append(px); allres.append(res); cams.append(cam.as_array())
        res = {k: np.concatenate([r[k] for r in allres]) for k in allres[0]}
        frame_of = np.concatenate([np.full(len(p), f, np.int32) for f, p in zip(frames, allpx)])
        meta = dict(width=1920, height=1080, samples=1, max_depth=5, mode="K", seed="frame",
                    frames=frames, n_frames=256, scene="generate_scene(8, seed 1, keep_default)")
        save(name, meta, skw, res, np.concatenate(allpx),
             extra=dict(frame_of=frame_of, cameras=np.array(cams, np.float32)))
        print(name, f"{time.time() - t0:.1f}s", flush=True)

    pool.close()
    print(f"done in {time.time() - t00:.1f}s")


if __name__ == "__main__":
    main()
