#!/usr/bin/env python3
"""rt355 benchmark: Mrays/s (primary + reflection + shadow) at 1080p, depth 5.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|t64|c3|c1|c4|c5|refr|cubes|cubes_glass]

One "step" = ``--batches`` (default 10) batches of ``--frames`` (default 1024, = one kernel launch pair) 1080p frames of
BASELINE.json config 2 (1920x1080, 8-sphere scene, depth 5, 13 shadow rays per shaded hit, keyed RNG, seed = frame
index), inputs (scene) already resident on the GPU: 10 240 frames, ~125 ms per step, so that the default 20 timed steps
last ~2.5 s and the clocks are at their steady state (a single batch takes 12 ms).  With N > 1 ranks (one process per GPU,
launched by torch.distributed.run) every frame is sharded by interleaved 8-row tiles
(tile t -> the (t mod G)-th of the G ranks that render: all N, or — in some batches of a fixed pattern, in all from five GPUs
on — ranks 1 .. N - 1 while GPU 0 places what has been gathered; RT_BENCH_LAYOUTS overrides), each rank renders its tiles of all frames of the batch, and ONE RCCL
gather per batch moves the occupied 8x8 tiles of the slabs (rt_compact_tiles: lossless, no host read-back, see
raytracer_amd.distributed.BatchExchange) to GPU 0, where rt_place_tiles writes the frames in image order:
total work per step is fixed => "strong" scaling.  ``--workload c5`` at N > 1 is BASELINE config 5's mode
instead: frame f -> rank f mod N, whole frames, no exchange inside the timed region; the one gather at the end
is timed separately (``final_gather_ms``).  Rays are counted by the kernels themselves (rt_get_ray_counts), not
modelled.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — FP32-VALU roofline of the launch pair rt_visibility_kernel + rt_stream_kernel (one
                 pair per batch: primary rays / everything after them).  `achieved` / `frac` are what the hardware
                 EXECUTED: FP32 flops the VALUs retired (per-ray PMC counts from the committed rocprofv3 pass named in
                 `pmc_profile`, scaled by this run's rays) over the pair's live HIP-event duration, against the 157.3
                 TFLOP/s peak — never above 1.  `issue_frac` = vector + scalar wave-instructions / issue slots.
                 `algorithmic_achieved` / `algorithmic_frac` price the flops of the reference's brute-force algorithm
                  over the same time: a speed-of-algorithm figure that exceeds 1 where the kernels cull
                 most of that work — it says how fast the ALGORITHM runs, not how busy the VALUs are;
  single_frame — the drop-in call itself, render(1920, 1080) -> numpy array (raytracer.py:310 renders ONE frame): scene
                 check, one launch pair, copy of the frame to the host; ms per call and the Mrays/s that makes;
  target_64sph — the same measurement, shorter, on the north-star target scene (1080p, depth 5, 64 spheres);
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
ISSUE_SLOTS_PER_S = 256 * 4 * 2.4e9 / 2   # a SIMD-32 issues one wave64 vector instruction per 2 clocks
PY_REFERENCE_MRAYS = 0.036        # the Python reference itself, 1 core (cannot run here)

# Per-launch-pair PMC sums of the committed rocprofv3 passes (tools/profile.sh + tools/summarize_profile.py), both
# kernels of a 1024-frame, 1-spp launch pair: FP32 flops the VALUs executed (SQ_INSTS_VALU_{ADD,MUL,TRANS}_F32 +
# 2 x FMA, x 64 lanes: a full-wave upper bound), vector and scalar wave-instructions, HBM bytes
# (TCC_EA0_WRREQ x 64 + TCC_EA0_RDREQ x 64 x 2, the guide's gfx950 correction), and the rays of that launch pair.
# (tools/pmc_constants.py prints these lines from a tools/profile.sh directory.)
PMC = {
    "c2": dict(profile="profiles/r3z_c2_rt_{stream,visibility}_kernel.txt", flops=3.64e+11, valu=7.719e+09, salu=2.929e+09, hbm_bytes=1.069e+10, rays=7.863e+09),
    "t64": dict(profile="profiles/r3z_t64_rt_{stream,visibility}_kernel.txt", flops=3.228e+11, valu=7.021e+09, salu=2.709e+09, hbm_bytes=1.164e+10, rays=6.428e+09),
    "c4": dict(profile="profiles/r3z_c4_rt_{stream,visibility}_kernel.txt", flops=2.724e+11, valu=5.175e+09, salu=1.348e+09, hbm_bytes=1.868e+10, rays=3.36e+09),
    "refr": dict(profile="profiles/r3z_refr_rt_{stream,visibility}_kernel.txt", flops=5.574e+11, valu=1.353e+10, salu=4.502e+09, hbm_bytes=1.144e+10, rays=9.619e+09),
    "cubes": dict(profile="profiles/r3z_cubes_rt_{stream,visibility}_kernel.txt", flops=7.309e+11, valu=1.769e+10, salu=5.734e+09, hbm_bytes=1.041e+10, rays=1.156e+10),
    "cubes_glass": dict(profile="profiles/r3z_cubes_glass_rt_{stream,visibility}_kernel.txt", flops=2.485e+12, valu=6.847e+10, salu=1.91e+10, hbm_bytes=2.45e+10, rays=3.152e+10),
}

WORKLOADS = {
    # name: (width, height, n_spheres, scene seed, keep_default, max_depth, description)
    "c1": (640, 480, 3, 0, True, 3, "config 1: default 3-sphere scene 640x480 depth 3"),
    "c2": (1920, 1080, 8, 1, True, 5, "config 2: 1920x1080, 8-sphere scene, depth 5, shadow rays"),
    "t64": (1920, 1080, 64, 2, False, 5, "north-star target: 1920x1080, 64 spheres, depth 5"),
    "c3": (3840, 2160, 64, 2, False, 8, "config 3: 3840x2160, 64 spheres, depth 8"),
    "c4": (7680, 4320, 1024, 3, False, 5, "config 4: 7680x4320, 1024 spheres, depth 5 (use --frames 16)"),
    "c5": (1920, 1080, 8, 1, True, 5, "config 5: 1080p camera-orbit animation over config 2's scene (use --frames 256)"),
    "s400": (1920, 1080, 400, 4, False, 5, "diagnostic: 1920x1080, 400 spheres, depth 5 (the list kernels below 513 spheres)"),
    "s700": (1920, 1080, 700, 5, False, 5, "diagnostic: 1920x1080, 700 spheres, depth 5 (the list kernels' eight-wave workgroups)"),
    "s2000": (1920, 1080, 2000, 6, False, 5, "diagnostic: 1920x1080, 2000 spheres, depth 5"),
    # SURVEY.md 8(f) rows measured beside the headline (not bench lines of BASELINE.json)
    "refr": (1920, 1080, 5, 0, True, 5, "8f rank 2: refractive_scene() (default scene + glass) 1920x1080 depth 5"),
    "cubes": (1920, 1080, 6, 0, True, 5, "8f rank 3: cube_scene() (default scene + 3 opaque cubes) 1920x1080 depth 5"),
    "cubes_glass": (1920, 1080, 6, 0, True, 5, "8f rank 3: cube_scene(glass=True) 1920x1080 depth 5"),
    "refr40": (1920, 1080, 40, 12, False, 5, "8f rank 2: 40 generated spheres, every third of glass, 1920x1080 depth 5"),
    "c2g": (1920, 1080, 8, 1, True, 5, "diagnostic: config 2 plus one glass sphere behind the camera, which no ray meets (the refractive kernels on config 2's rays)"),
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
    scene = default_scene() if n == 3 else generate_scene(n, sseed, w, h, keep_default=keep)
    if wl == "c2g":
        from raytracer_amd.scenes import Material, Sphere, Vector3
        scene["spheres"].append(Sphere(Vector3(0.0, 0.0, 60.0), 1.0, Material(Vector3(1, 1, 1), specular=0.5, reflection=0.0,
                                                                              refraction=0.5, refractive_index=1.5)))
    if wl == "refr40":
        for i, s_ in enumerate(scene["spheres"]):
            if i % 3 == 0:
                s_.material.refraction, s_.material.refractive_index = 0.6, 1.4
    return scene


def cpu_baseline(wl, target_s=5.0):
    """Time the oracle (FP64 restatement of the reference) on this host, all cores, on the same workload until
    ~target_s seconds have been spent.  A sample "frame" is the whole frame where one costs about what a config-2
    frame does; bigger workloads (more pixels x spheres x depth) are sampled on a regular pixel lattice of the
    frame (every k-th pixel of every k-th row), so the ray mix stays the workload's and the run stays bounded."""
    from oracle import cpu_oracle
    import numpy as np
    w, h, n, sseed, keep, D, _ = WORKLOADS[wl]
    scene = build_scene(wl)
    threads = min(cpu_oracle.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 10**6,
                  len(os.sched_getaffinity(0)))

    from raytracer_amd.scenes import pack_scene
    p = pack_scene(scene)
    cost = (w * h) * max(p.n_spheres, 8) / (1920.0 * 1080.0 * 8.0)
    k = max(1, int(np.ceil(np.sqrt(cost))))
    pixels = None
    if k > 1:
        ys, xs = np.meshgrid(np.arange(k // 2, h, k), np.arange(k // 2, w, k), indexing="ij")
        pixels = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)

    def run_frame(seed, nthreads=None):
        t0 = time.perf_counter()
        o = cpu_oracle.render_scene(p, w, h, pixels=pixels, max_depth=D, mode="K", seed=seed,
                                    threads=threads if nthreads is None else nthreads)
        dt = time.perf_counter() - t0
        rays = int(o["trace_levels"].sum()) + (1 + p.area_samples) * int(o["hit_levels"].sum())
        return rays, dt

    # sample frames of the workload (seed = frame index) until ~target_s of CPU time is spent
    rays = frames = 0
    dt = 0.0
    while dt < target_s and frames < 1024:
        r_, d_ = run_frame(frames)
        rays += r_; dt += d_; frames += 1
    r1, d1 = run_frame(0, 1)     # the same restatement on one thread, one sample frame
    what = "whole frames of the workload" if pixels is None else \
        f"sample frames of the workload (every {k}th pixel of every {k}th row: {len(pixels)} pixels each)"
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample=f"{frames} {what} ({rays} rays, {dt:.1f} s), FP64, OpenMP",
                single_thread_mrays=r1 / d1 / 1e6,
                python_reference_mrays_1core=PY_REFERENCE_MRAYS)


def single_frame(r, wl):
    """The reference's API renders ONE frame (raytracer.py:310): time the drop-in call — scene check (an unchanged scene is
    not uploaded again), one launch pair for one frame, copy of the frame to a pinned host array."""
    import raytracer_amd
    render_mod = sys.modules["raytracer_amd.render"]       # (the package exports the FUNCTION render under that name)
    w, h, _, _, _, D, _ = WORKLOADS[wl]
    scene = build_scene(wl)
    saved = render_mod.scene
    render_mod.scene = scene
    try:
        for i in range(5):
            raytracer_amd.render(w, h, 1, max_depth=D, seed=i)
        rr = render_mod._renderer(None)
        rr.reset_ray_counts()
        n = 50
        t0 = time.perf_counter()
        for i in range(n):
            img = raytracer_amd.render(w, h, 1, max_depth=D, seed=i)
        dt = time.perf_counter() - t0
        rays = rr.ray_counts()["total"]
    finally:
        render_mod.scene = saved
    assert img.shape == (h, w, 3)
    return {"ms_per_call": dt / n * 1e3, "mrays_per_s": rays / dt / 1e6, "calls": n,
            "what": f"raytracer_amd.render({w}, {h}) -> numpy uint8 [H, W, 3] on the host, one call after another "
                    "(scene check, one launch pair, device-to-host copy of the frame)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=1024, help="frames per batch (one kernel launch pair)")
    ap.add_argument("--batches", type=int, default=10, help="batches per step")
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--kernel", default="stream", choices=["stream", "pixel"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target", action="store_true", help="skip the second timed region on the 64-sphere target scene")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the timing of the one-frame drop-in call")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
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
    # staged through host memory); the driver's runs use nccl = RCCL, one rank per GPU.
    # RT_BENCH_FORCE_EXCHANGE=1 at N = 1: the whole N > 1 step — compaction, count exchange, gather, placement, every
    # collective through RCCL — as a group of ONE rank (the rehearsal a one-GPU box allows).
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    force_exchange = bool(os.environ.get("RT_BENCH_FORCE_EXCHANGE"))
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    from raytracer_amd import tiling
    from raytracer_amd._lib import FRAMES_PER_LAUNCH
    from raytracer_amd.distributed import BatchExchange, frames_of_rank, gather_frames
    from raytracer_amd.render import Renderer
    from raytracer_amd.scenes import Camera, orbit_camera, pack_scene

    dev = f"cuda:{local_rank}"
    red_dev = "cpu" if backend == "gloo" else dev
    r = Renderer(local_rank)
    r.set_kernel(args.kernel)
    if os.environ.get("RT_BENCH_NO_SPILL"):     # A/B of the refractive queue's spill stack
        r.set_queue_spill(False)
    if os.environ.get("RT_BENCH_NO_TILE_FLUSH"):  # A/B of the stream kernel's coalesced output path
        r.set_tile_flush(False)
    buffers = {}                                # frame / exchange buffers, reused by workloads of the same size
    # The collectives of the N > 1 step run inside librt355 (rt_comm_*: the library's own RCCL communicator, SURVEY.md
    # 5 / 8b) unless RT_BENCH_COMM=torch asks for torch.distributed's; torch.distributed stays for the rendezvous, the
    # barrier and the timing reduction.  The communicator is tried once on known values before any timed work; if a rank
    # could not join or a value came out wrong, every rank falls back to torch.distributed's collectives and the JSON
    # line says so.
    comm, comm_note = None, "none (single GPU)"
    if world > 1 or force_exchange:
        comm_note = "torch.distributed (backend %s)" % backend
        if backend == "nccl" and os.environ.get("RT_BENCH_COMM", "abi") == "abi":
            ok = 0
            try:
                r.comm_init()
                one = torch.tensor([rank + 1], dtype=torch.int32, device=dev)
                every = torch.zeros(world, dtype=torch.int32, device=dev)
                r.comm_all_gather_counts(one, every)
                send = torch.full(((rank + 1) * 196,), rank + 1, dtype=torch.uint8, device=dev)
                got = torch.zeros((world, world * 196), dtype=torch.uint8, device=dev) if rank == 0 else None
                r.comm_gather(send, (rank + 1) * 196, got, [(k + 1) * 196 for k in range(world)], world * 196, 0)
                torch.cuda.synchronize()
                ok = int(every.tolist() == list(range(1, world + 1)) and
                         (rank != 0 or all(bool((got[k, :(k + 1) * 196] == k + 1).all()) and not bool(got[k, (k + 1) * 196:].any())
                                           for k in range(world))))
            except Exception as e:      # noqa: BLE001  (reported; the fallback below takes over)
                print(f"[bench] rank {rank}: rt_comm_* unavailable: {e!r}", file=sys.stderr, flush=True)
            agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 1:
                comm = r
                v = r.comm_info()["rccl_version"]
                comm_note = f"librt355 rt_comm_* (grouped ncclSend / ncclRecv, RCCL {v // 10000}.{v // 100 % 100}.{v % 100} bound at run time)"
            else:
                comm_note += " — rt_comm_* failed its self-test on some rank"

    # Row tiles at N > 1: which ranks render.  GPU 0 writes every byte of every frame (rt_place_tiles: 0.56 ms per 256 1080p
    # frames whatever N) and that does not hide behind its own render — the shading kernel fills the register file, a
    # kernel of another stream runs behind it, not beside it (tools/root_emulation.py) — so a root with a full share is the
    # slowest rank from N = 3 on.  Two layouts, one letter per batch of a repeating pattern:
    #   A — all N ranks render, tile t -> rank t mod N;      B — ranks 1 .. N - 1 render, tile t -> rank 1 + t mod (N - 1),
    #                                                            the root gathers and places only.
    # The layouts take turns in a pattern that gives the root the share of rendering its placements leave room for: in an A
    # batch every rank renders the same share and the root places NOTHING; at the start of each B batch the root places
    # everything gathered so far on its render stream (BatchExchange(defer_place=True)) while the others render the B batch —
    # about as long, by the pieces tools/scaling_model.py measures: AAAAAB at N = 2, AAB at 3, AB at 4, B alone from 5 on.
    # RT_BENCH_LAYOUTS=<letters> overrides ("A" / "B": one layout throughout).
    pattern = {1: "A", 2: "AAAAAB", 3: "AAB", 4: "AB"}.get(world, "B")
    if os.environ.get("RT_BENCH_ROOT_RENDERS") in ("0", "1"):
        pattern = "A" if os.environ["RT_BENCH_ROOT_RENDERS"] == "1" else "B"
    if os.environ.get("RT_BENCH_LAYOUTS"):
        pattern = os.environ["RT_BENCH_LAYOUTS"].upper()
    if world == 1 or set(pattern) - {"A", "B"} or not pattern:
        pattern = "A"

    def measure(wl, F, B, steps, warmup, samples):
        """warmup + `steps` timed steps (of B batches of F frames) of workload `wl`; returns the numbers of the JSON line
        (rank 0) or None."""
        w, h, n_sph, _, _, D, desc = WORKLOADS[wl]
        tr = args.tile_rows
        packed = pack_scene(build_scene(wl))
        r.upload_scene(packed)
        frame_parallel = wl == "c5" and world > 1           # BASELINE config 5: frame f -> rank f mod N, whole frames
        use_ex = (world > 1 or force_exchange) and not frame_parallel
        letters = sorted(set(pattern)) if use_ex else ["A"]

        def layout(letter):     # ranks that render row tiles, this rank's place among them (-1: none), its rows
            G = world if (letter == "A" or frame_parallel) else world - 1
            ridx = rank if G == world else rank - 1
            return G, ridx, (h if frame_parallel else tiling.max_local_rows(h, tr, G))
        key = (w, h, F, frame_parallel, tuple(letters))
        if buffers.get("key") != key:
            buffers.clear()
            buffers["key"] = key
            buffers["lay"] = {}
            # N > 1: every rank that renders puts its row tiles of the F frames of a batch into a slab; one gather per batch
            # moves the occupied tiles of the slabs to GPU 0, where one pass writes the frames in image order.
            # Pipelined: compaction, gather and placement of earlier batches overlap the render of later ones, and no
            # rank's host waits for its own render (BatchExchange).  RT_BENCH_DENSE=1 sends whole slabs.  Each layout has
            # its own exchange (slabs of its size) and its own frames on the root.
            for letter in letters:
                G, ridx, rows_max = layout(letter)
                L = dict(G=G, ridx=ridx, rows_max=rows_max, ex=None, fo=None, count=0)
                if rank == 0 or frame_parallel:
                    n_local = len(frames_of_rank(F, rank, world)) if frame_parallel else F
                    L["fo"] = torch.empty((n_local, h, w, 3), dtype=torch.uint8, device=dev)
                if use_ex:
                    slab_bytes = rows_max * w * 3
                    if os.environ.get("RT_BENCH_DENSE"):
                        L["ex"] = BatchExchange(F, slab_bytes, dev,
                                                lambda gathered, i, L=L: r.scatter_rows_frames(gathered[world - L["G"]:], L["G"], w, h, tr, out=L["fo"]))
                    else:
                        def place_tiles(entries, counts, i, L=L, table={}):
                            # (called with the stream current on which the placement is to run: the exchange's side stream, or —
                            # where the root renders in this layout — its render stream; a root that does not render has no
                            # entries: the renderers' are ranks 1 .. N - 1 of the gather)
                            table["t"] = r.place_tiles(entries[world - L["G"]:], counts[world - L["G"]:], w, h, tr, L["fo"], table.get("t"),
                                                       on_current_stream=L["ex"].on_side)
                        L["ex"] = BatchExchange(F, slab_bytes, dev, None, compact=r.compact_tiles, place_tiles=place_tiles,
                                                capacity=F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8),
                                                force_collectives=force_exchange, comm=comm, contributes=ridx >= 0,
                                                side_stream=buffers.get("side"),
                                                # (layouts taking turns: the root places when it is idle)
                                                defer_place=len(letters) > 1)
                if L["ex"] is not None and getattr(L["ex"], "on_side", False):
                    buffers["side"] = L["ex"].side          # (one side stream for every layout's collectives)
                buffers["lay"][letter] = L
        lay = buffers["lay"]
        exs = [L["ex"] for L in lay.values() if L["ex"] is not None]
        seq = pattern if use_ex else "A"
        # (averages over the pattern, for the per-launch figures below)
        rows_max = max(L["rows_max"] for L in lay.values())
        rows_mean = sum(lay[c]["rows_max"] for c in seq) / len(seq)
        renders_here = any(L["ridx"] >= 0 for L in lay.values())
        render_share = sum(1 for c in seq if lay[c]["ridx"] >= 0) / len(seq)     # of the batches, on this rank
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        cams = [Camera()] * F                       # config 2's fixed pinhole; frames differ by seed
        if wl == "c5":                              # BASELINE config 5: frame f of a 256-frame orbit
            cams = [orbit_camera(f % 256, 256) for f in range(F)]
        mine = frames_of_rank(F, rank, world) if frame_parallel else None
        if frame_parallel:
            cams = [cams[int(f)] for f in mine]
        cams = r.pack_cameras(cams)                 # packed once: the step loop must not be host-bound at N = 8

        def step(i, timed_idx=None):
            for b in range(B):                      # batch number j: its frames are j F .. (j + 1) F - 1 of the run
                j = i * B + b
                L = lay[seq[j % len(seq)]]
                ex = L["ex"]
                if timed_idx is not None and b == 0:
                    ev[timed_idx][0].record()
                k = L["count"]                      # (each layout's exchange counts its own batches)
                L["count"] += 1
                if L["ridx"] < 0:                   # (layout B on the root: it gathers and places)
                    # layouts taking turns: this is the root's idle batch — everything gathered so far is placed now, on the
                    # render stream, which has nothing else to do until the next A batch
                    for e in exs:
                        e.place_pending()
                elif frame_parallel:                # this rank's frames of the batch, whole
                    r.render_frames(cams, [j * F + int(f) for f in mine], w, h, samples, max_depth=D, rng_mode="K", out=L["fo"])
                else:
                    # N = 1: the rank owns every row, so its packed slab IS the frame batch
                    out = L["fo"] if ex is None else ex.slab(k).view(F, L["rows_max"], w, 3)
                    r.render_frames(cams, range(j * F, (j + 1) * F), w, h, samples, max_depth=D, rng_mode="K",
                                    out=out, tile_rows=tr, tile_first=L["ridx"], tile_stride=L["G"])
                if timed_idx is not None and b == B - 1:
                    ev[timed_idx][1].record()
                if ex is not None:
                    ex.submit(k)

        def barrier():
            for ex in exs:
                ex.drain()         # the last batches' gathers and placements belong to the timed region
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(warmup):
            step(i)
        barrier()
        r.reset_ray_counts()
        for ex in exs:
            ex.sent_bytes = 0
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, i)
        barrier()
        dt = time.perf_counter() - t0

        gather_ms = None
        if frame_parallel:                          # config 5: the one exchange, after the clock has stopped
            t1 = time.perf_counter()
            gather_frames(lay["A"]["fo"], F)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - t1) * 1e3

        rc = r.ray_counts()
        stats = torch.tensor([rc["total"], rc["shaded_hits"], rc["primary"]], dtype=torch.float64, device=red_dev)
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        # (refractive scenes whose pixel sums lie in device memory — several samples per pixel —: as many frames per launch
        # pair as fit 32 GiB of them, kRefrSumBytes in csrc/rt355.hip)
        refr_slabs = packed.has_refraction and (samples > 64 or os.environ.get("RT_BENCH_NO_TILE_FLUSH"))
        per_launch = min(FRAMES_PER_LAUNCH, max(1, (32 << 30) // (12 * rows_max * w))) if refr_slabs else FRAMES_PER_LAUNCH
        n_local_frames = len(mine) if frame_parallel else F
        one_stream_launch = samples == 1 or (2 <= samples <= 64 and args.kernel == "stream"
                                             and not os.environ.get("RT_BENCH_NO_TILE_FLUSH"))
        launches_per_step = B * -(-n_local_frames // per_launch) * (1 if one_stream_launch else samples)
        # (with several batches per step the events also cover the other stream-ordered work between the pairs: the
        # exchange's compaction kernels at N > 1)
        # (a rank that sits out some batches of the pattern — the root — reports no launch time: its events span the batches it
        # only places; the busiest rank's is taken below)
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / (steps * launches_per_step) if render_share == 1.0 else 0.0
        # (launch time, rays and shaded hits of the busiest rank: what the per-launch figures of the roofline block are about —
        # rank 0's own where it renders alone)
        kmax = torch.tensor([kern_ms, rc["total"], rc["shaded_hits"]], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        if rank != 0:
            return None
        kern_ms, rank_rays, rank_shaded = (float(x) for x in kmax.tolist())
        total_rays, shaded, primary = (float(x) for x in stats.tolist())
        dt = float(tmax.item())
        launches = steps * launches_per_step
        secs = kern_ms * 1e-3
        # per-launch (per rank) algorithmic flops of the launch pair on rank 0
        fl = algorithmic_flops(rank_rays, rank_shaded, n_sph, packed.area_samples) / launches
        roof = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "frac_is": "EXECUTED: FP32 flops the VALUs retired (PMC SQ_INSTS_VALU_{ADD,MUL,TRANS}_F32 + 2 x FMA_F32, x 64 lanes: "
                           "a full-wave upper bound; per-ray counts of the committed profile `pmc_profile` of this workload, scaled "
                           "by this run's rays) / this run's launch-pair time / 157.3 TFLOP/s.  null where no profile of the "
                           "workload is committed",
                "traffic": None, "issue_frac": None, "pmc_profile": None,
                "kernel": ("rt_visibility_kernel + rt_stream_kernel (one pair per launch; the events also cover the ~4 us "
                           "rt_camera_table_kernel, up to 16 frame-record stores and a 4 MB memset)")
                if args.kernel == "stream" else "rt_pixel_kernel",
                "kernel_ms": kern_ms, "launches_per_step": launches_per_step,
                "algorithmic_flops_per_launch": fl, "algorithmic_achieved": fl / secs / 1e12,
                "algorithmic_frac": fl / secs / 1e12 / PEAK_FP32_TFLOPS,
                "algorithmic_is": "flops of the reference's brute-force algorithm (rays*(5+17N) + shaded*(55+75*(1+ns)), SURVEY.md "
                                  "8d) / time / peak: a speed-of-algorithm figure, above 1 where the kernels cull most of that work"}
        pmc = PMC.get(wl)
        if pmc and samples == 1 and args.kernel == "stream" and not os.environ.get("RT_BENCH_NO_TILE_FLUSH"):
            # what the hardware executed: the profiled launch pair's counts per ray x this run's rays per launch
            scale = (rank_rays / launches) / pmc["rays"]
            roof["achieved"] = pmc["flops"] * scale / secs / 1e12
            roof["frac"] = roof["achieved"] / PEAK_FP32_TFLOPS
            roof["executed_flops_per_launch"] = pmc["flops"] * scale
            roof["issue_frac"] = (pmc["valu"] + pmc["salu"]) * scale / secs / ISSUE_SLOTS_PER_S
            roof["valu_issue_frac"] = pmc["valu"] * scale / secs / ISSUE_SLOTS_PER_S
            roof["traffic"] = pmc["hbm_bytes"] * scale
            roof["frame_bytes_per_launch"] = float(n_local_frames) * rows_mean * w * 3
            roof["pmc_profile"] = pmc["profile"]
            roof["pmc_note"] = ("issue_frac = (vector + scalar wave-instructions) / (1024 SIMDs x 1.2e9 issue slots/s: one wave64 "
                                "instruction per 2 clocks); traffic = HBM bytes per launch pair (TCC_EA0_*); PMC counts per ray from "
                                "the named profile of this workload, scaled by this run's rays")
        if frame_parallel:
            par = f"frame-parallel x{world}: frame f -> rank f mod {world}, whole frames, one gather after the timed region"
        elif exs:
            kinds = {"A": f"A = all {world} ranks render, tile t -> rank t mod {world}",
                     "B": f"B = ranks 1..{world - 1} render, tile t -> rank 1 + t mod {world - 1}, GPU 0 gathers and places only"}
            par = f"row tiles of {tr} rows, layouts by batch in the pattern {seq!r} (" + "; ".join(kinds[c] for c in sorted(set(seq))) + \
                  ") + RCCL gather to GPU 0" + (" of the occupied 8x8 tiles" if exs[0].sparse else "") + \
                  (" (a group of ONE rank: rehearsal of the N > 1 step)" if world == 1 else "")
        else:
            par = "single GPU"
        out = {
            "value": total_rays / dt / 1e6,
            "ms_per_step": dt / steps * 1e3,
            "config": {"workload": f"{desc}; step = {B} batches of {F} frames (one launch pair each), {samples} spp, "
                                   f"keyed RNG, {n_sph} spheres, 12 area-light samples",
                       "width": w, "height": h, "max_depth": D, "frames_per_batch": F, "batches_per_step": B,
                       "frames_per_step": F * B, "parallelism": par, "collectives": comm_note},
            "rays_per_frame": total_rays / (steps * F * B),
            "rays_per_pixel": total_rays / max(primary, 1.0),
            "mpixels_per_s": primary / dt / 1e6,
            "frames_per_s": steps * F * B / dt,
            "timed_region_s": dt,
            "exchange_bytes_per_step_rank0": sum(ex.sent_bytes for ex in exs) / steps,
            "roofline": roof,
        }
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
            out["final_gather_note"] = (f"one gather of the last batch's {F} frames to GPU 0 after the timed region; "
                                        "frames_per_s is at render completion (SURVEY.md 8e)")
        return out

    # With an exchange, the render stream is a stream of its own, not HIP's legacy default stream (torch's current stream unless
    # told otherwise): the default stream synchronises implicitly with every blocking stream of the process, and a default
    # stream that waits for an event behind the library's ncclSend / ncclRecv deadlocked in the one-GPU rehearsal (deferred
    # placement on the library's communicator) where an ordinary stream does not.
    import contextlib
    render_stream = torch.cuda.Stream(device=dev) if (world > 1 or force_exchange) else None
    with (torch.cuda.stream(render_stream) if render_stream is not None else contextlib.nullcontext()):
        r.use_torch_stream()
        res = measure(args.workload, args.frames, args.batches, args.steps, args.warmup, args.samples)
        target = None
        if not args.no_target and args.workload != "t64":
            # the north-star target scene (1080p, depth 5, 64 spheres): the same step, the same number of steps
            target = measure("t64", args.frames, args.batches, args.steps, 1, 1)
    r.use_torch_stream()
    single = None
    if rank == 0 and world == 1 and not args.no_single_frame and args.kernel == "stream":
        single = single_frame(r, args.workload)

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+secondary) at 1080p depth 5",
            "value": res["value"],
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
        }
        out.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step")})
        if target is not None:
            tr_ = target["roofline"]
            out["target_64sph"] = {"value": target["value"], "unit": "Mrays/s", "ms_per_step": target["ms_per_step"],
                                   "workload": target["config"]["workload"], "kernel_ms": tr_["kernel_ms"],
                                   "frac": tr_["frac"], "algorithmic_frac": tr_["algorithmic_frac"],
                                   "issue_frac": tr_["issue_frac"], "pmc_profile": tr_["pmc_profile"],
                                   "frames_per_s": target["frames_per_s"], "timed_region_s": target["timed_region_s"]}
        if single is not None:
            out["single_frame"] = single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
