#!/usr/bin/env python3
"""rt355 benchmark: Mrays/s (primary + reflection + shadow) at 1080p, depth 5.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|t64|c3|c1]

One "step" = one batch of ``--frames`` (default 1024, = one kernel launch pair) 1080p frames of BASELINE.json config 2
(1920x1080, 8-sphere scene, depth 5, 13 shadow rays per shaded hit, keyed RNG, seed = frame
index), inputs (scene) already resident on the GPU.  With N > 1 ranks (one process per GPU,
launched by torch.distributed.run) every frame is sharded by interleaved 8-row tiles
(tile t -> rank t mod N), each rank renders its tiles of all frames of the batch, and ONE RCCL
gather per step moves the packed tiles (only their non-black 32-byte segments: lossless, see
raytracer_amd.distributed.BatchExchange) to GPU 0, where rt_scatter_rows puts them in image order:
total work per step is fixed => "strong" scaling.  Rays are counted by the kernel itself
(rt_get_ray_counts), not modelled.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — FP32-VALU roofline of the launch pair rt_visibility_kernel + rt_stream_kernel (one
                 pair per batch and sample: primary rays / everything after them) from algorithmic
                 flops and the pair's live HIP-event duration on its launch stream;
  cpu_baseline — the FP64 C++ restatement of the reference (oracle/, "port") timed on this
                 host's cores on a bounded number of whole frames of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
MEASURED_TRAFFIC_C2 = 3.26e10       # bytes per 1024-frame launch pair, profiles/r1m_c2_* (see roofline.traffic)
PY_REFERENCE_MRAYS = 0.036        # the Python reference itself, 1 core (cannot run here)

WORKLOADS = {
    # name: (width, height, n_spheres, scene seed, keep_default, max_depth, description)
    "c1": (640, 480, 3, 0, True, 3, "config 1: default 3-sphere scene 640x480 depth 3"),
    "c2": (1920, 1080, 8, 1, True, 5, "config 2: 1920x1080, 8-sphere scene, depth 5, shadow rays"),
    "t64": (1920, 1080, 64, 2, False, 5, "north-star target: 1920x1080, 64 spheres, depth 5"),
    "c3": (3840, 2160, 64, 2, False, 8, "config 3: 3840x2160, 64 spheres, depth 8"),
    "c4": (7680, 4320, 1024, 3, False, 5, "config 4: 7680x4320, 1024 spheres, depth 5 (use --frames 16)"),
    "c5": (1920, 1080, 8, 1, True, 5, "config 5: 1080p camera-orbit animation over config 2's scene (use --frames 256)"),
    # SURVEY.md 8(f) rows measured beside the headline (not bench lines of BASELINE.json)
    "refr": (1920, 1080, 5, 0, True, 5, "8f rank 2: refractive_scene() (default scene + glass) 1920x1080 depth 5"),
    "cubes": (1920, 1080, 6, 0, True, 5, "8f rank 3: cube_scene() (default scene + 3 opaque cubes) 1920x1080 depth 5"),
    "cubes_glass": (1920, 1080, 6, 0, True, 5, "8f rank 3: cube_scene(glass=True) 1920x1080 depth 5"),
}


def algorithmic_flops(rays: int, shaded: int, n_spheres: int, area_samples: int) -> float:
    """SURVEY.md §8d, conservative: 5 + 17*N per ray (the 3 flops per candidate root are not
    counted), plus per shaded hit 30 (point, normal) + (1 + samples) * 75 (light dir, offset origin,
    diffuse, R, V.R, pow-50, accumulate) + 25 (reflect + mix)."""
    return rays * (5.0 + 17.0 * n_spheres) + shaded * (30.0 + (1 + area_samples) * 75.0 + 25.0)


def build_scene(wl):
    from raytracer_amd.scenes import cube_scene, default_scene, generate_scene, refractive_scene
    w, h, n, sseed, keep, D, _ = WORKLOADS[wl]
    if wl in ("refr", "cubes", "cubes_glass"):
        return refractive_scene() if wl == "refr" else cube_scene(glass=(wl == "cubes_glass"))
    return default_scene() if n == 3 else generate_scene(n, sseed, w, h, keep_default=keep)


def cpu_baseline(wl, target_s=15.0):
    """Time the oracle (FP64 restatement of the reference) on this host, all cores, on whole frames
    of the same workload until ~target_s seconds of CPU time have been spent."""
    import numpy as np
    from oracle import cpu_oracle
    w, h, n, sseed, keep, D, _ = WORKLOADS[wl]
    scene = build_scene(wl)
    threads = min(cpu_oracle.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 10**6,
                  len(os.sched_getaffinity(0)))

    from raytracer_amd.scenes import pack_scene
    p = pack_scene(scene)

    def run_frame(seed, nthreads=None):
        t0 = time.perf_counter()
        o = cpu_oracle.render_scene(p, w, h, max_depth=D, mode="K", seed=seed,
                                    threads=threads if nthreads is None else nthreads)
        dt = time.perf_counter() - t0
        rays = int(o["trace_levels"].sum()) + (1 + p.area_samples) * int(o["hit_levels"].sum())
        return rays, dt

    # whole frames of the workload (seed = frame index) until ~target_s of CPU time is spent
    rays = frames = 0
    dt = 0.0
    while dt < target_s and frames < 1024:
        r_, d_ = run_frame(frames)
        rays += r_; dt += d_; frames += 1
    r1, d1 = run_frame(0, 1)     # the same restatement on one thread, one frame
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample=f"{frames} whole frames of the workload ({rays} rays, {dt:.1f} s), FP64, OpenMP",
                single_thread_mrays=r1 / d1 / 1e6,
                python_reference_mrays_1core=PY_REFERENCE_MRAYS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=1024, help="frames per step (batch)")
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--kernel", default="stream", choices=["stream", "pixel"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: rt355 has no CPU path")
    # RT_BENCH_BACKEND=gloo rehearses the N > 1 code path with all ranks on the GPUs that exist (slabs are
    # staged through host memory); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from raytracer_amd.distributed import BatchExchange
    from raytracer_amd.render import Renderer
    from raytracer_amd.scenes import pack_scene
    from raytracer_amd import tiling

    w, h, n_sph, _, _, D, desc = WORKLOADS[args.workload]
    F, tr = args.frames, args.tile_rows
    scene = build_scene(args.workload)
    packed = pack_scene(scene)
    r = Renderer(local_rank)
    r.upload_scene(packed)
    r.set_kernel(args.kernel)
    if os.environ.get("RT_BENCH_NO_SPILL"):     # A/B of the refractive queue's spill stack
        r.set_queue_spill(False)
    if os.environ.get("RT_BENCH_NO_TILE_FLUSH"):  # A/B of the stream kernel's coalesced output path
        r.set_tile_flush(False)

    rows_max = tiling.max_local_rows(h, tr, world)
    slab_bytes = rows_max * w * 3
    dev = f"cuda:{local_rank}"
    if rank == 0:
        frames_out = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
    # N > 1: every rank renders its row tiles of the F frames of a step into a slab; one gather per step
    # moves the slabs to GPU 0, where one launch puts the rows in image order.  Double-buffered: the
    # gather of step i (RCCL's stream) overlaps the render of step i + 1.
    # Only the 32-byte segments of a slab that are not black travel (lossless, BatchExchange(sparse=True)): the
    # exchange, not the rays, bounds N > 1.  RT_BENCH_DENSE=1 sends whole slabs.
    # The root puts gathered segments straight into image order (segments_to_frames: no dense gather buffer).
    from raytracer_amd.distributed import segments_to_frames
    direct = (w * 3) % 32 == 0 and not os.environ.get("RT_BENCH_DENSE_PLACE")
    ex = BatchExchange(F, slab_bytes, dev,
                       lambda gathered, i: r.scatter_rows_frames(gathered, world, w, h, tr, out=frames_out),
                       sparse=not os.environ.get("RT_BENCH_DENSE"),
                       place_segments=(lambda rows, counts, i: segments_to_frames(rows, counts, frames_out, w, h, tr, rows_max))
                       if direct else None) \
        if (world > 1 or os.environ.get("RT_BENCH_FORCE_EXCHANGE")) else None   # (env: rehearse this path on 1 GPU)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    from raytracer_amd.scenes import Camera
    cams = [Camera()] * F                       # config 2's fixed pinhole; frames differ by seed
    if args.workload == "c5":                   # BASELINE config 5: frame f of a 256-frame orbit
        from raytracer_amd.scenes import orbit_camera
        cams = [orbit_camera(f % 256, 256) for f in range(F)]
    cams = r.pack_cameras(cams)                 # packed once: the step loop must not be host-bound at N = 8

    def step(i, timed_idx=None):
        # N = 1: the rank owns every row, so its packed slab IS the frame batch
        out = frames_out if ex is None else ex.slab(i).view(F, rows_max, w, 3)
        if timed_idx is not None:
            ev[timed_idx][0].record()
        r.render_frames(cams, range(i * F, (i + 1) * F), w, h, args.samples, max_depth=D, rng_mode="K",
                        out=out, tile_rows=tr, tile_first=rank, tile_stride=world)
        if timed_idx is not None:
            ev[timed_idx][1].record()
        if ex is not None:
            ex.submit(i)

    def barrier():
        if ex is not None:
            ex.drain()         # the last steps' gathers and placements belong to the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    r.reset_ray_counts()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, i)
    barrier()
    dt = time.perf_counter() - t0

    rc = r.ray_counts()
    red_dev = "cpu" if backend == "gloo" else dev
    stats = torch.tensor([rc["total"], rc["shaded_hits"], rc["primary"]], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    from raytracer_amd._lib import FRAMES_PER_LAUNCH
    # (refractive scenes: as many frames per launch pair as fit 32 GiB of float pixel sums, kRefrSumBytes in csrc/rt355.hip)
    per_launch = min(FRAMES_PER_LAUNCH, max(1, (32 << 30) // (12 * rows_max * w))) if packed.has_refraction else FRAMES_PER_LAUNCH
    launches_per_step = -(-F // per_launch) * args.samples
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / (args.steps * launches_per_step)
    kmax = torch.tensor([kern_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    total_rays, shaded, primary = (float(x) for x in stats.tolist())
    dt = float(tmax.item())

    if rank == 0:
        # per-launch (per rank) algorithmic flops of rt_render_kernel on rank 0
        launches = args.steps * launches_per_step
        fl = algorithmic_flops(rc["total"], rc["shaded_hits"], n_sph, packed.area_samples) / launches
        achieved = fl / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/s (primary+secondary) at 1080p depth 5",
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{desc}; batch of {F} frames per step, {args.samples} spp, "
                                   f"keyed RNG, {n_sph} spheres, 12 area-light samples",
                       "width": w, "height": h, "max_depth": D, "frames_per_step": F,
                       "parallelism": f"row-tiles x{world} (tile {tr} rows) + RCCL gather to GPU 0"
                                      + (" of the non-black 32-byte segments" if ex is not None and ex.sparse else "")
                       if world > 1 else "single GPU"},
            "rays_per_frame": total_rays / (args.steps * F),
            "rays_per_pixel": total_rays / max(primary, 1.0),
            "mpixels_per_s": primary / dt / 1e6,
            "frames_per_s": args.steps * F / dt,
            "exchange_bytes_per_step_rank0": (ex.sent_bytes / (args.steps + args.warmup)) if ex is not None else 0,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         # HBM bytes per launch pair: a separate rocprofv3 --pmc pass of this same command
                         # (tools/profile.sh; TCC_EA0_WRREQ_sum x 64 B + TCC_EA0_RDREQ_sum x 64 B x 2, the
                         # gfx950 correction of MI355X_MICROARCH.md), profiles/r1m_c2_*: visibility 7.7e9 W,
                         # stream 2.37e10 W + 1.1e9 R.  Known for the default workload only.
                         "traffic": MEASURED_TRAFFIC_C2 if (args.workload == "c2" and F == 1024 and args.samples == 1
                                                            and world == 1 and args.kernel == "stream") else None,
                         "kernel": ("rt_visibility_kernel + rt_stream_kernel (one pair per launch; the events "
                                    "also cover the ~4 us rt_camera_table_kernel, up to 16 frame-record stores and a 4 MB memset)")
                         if args.kernel == "stream" else "rt_pixel_kernel", "kernel_ms": kern_ms,
                         "flops_per_launch": fl,
                         "note": "algorithmic flops = rays*(5+17N) + shaded*(55+75*(1+ns)), SURVEY.md 8d, i.e. "
                                 "the brute-force count of the reference's algorithm: the kernels cull spheres (per object, per "
                                 "shading point, per ray bundle) and execute ~3x fewer flops, so frac exceeds 1 on "
                                 "many-sphere scenes (DESIGN.md 4, roofline); HBM traffic (PMC, profiles/r1m_*) is "
                                 "~3.3e10 B per 1024-frame config-2 launch: the framebuffer, 2.25 B/pixel of hit indices and the "
                                 "scattered pixel stores"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
