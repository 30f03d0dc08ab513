#!/usr/bin/env python3
"""Phase breakdown of the stream kernel from a -DRT_STAMPS=1 build (diagnostic; never quote its
run time).  RT355_LIB=<stamps build> python tools/stamps.py [workload] [frames]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera

wl = sys.argv[1] if len(sys.argv) > 1 else "t64"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
SPP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h, n, _, _, D, desc = bench.WORKLOADS[wl]
r = Renderer(0)
r.upload_scene(bench.build_scene(wl))
if os.environ.get("RT_BENCH_NO_TILE_FLUSH"):
    r.set_tile_flush(False)
for it in range(2):
    r.reset_ray_counts()
    r.render_frames([Camera()] * F, range(F), w, h, SPP, max_depth=D)
a = (ctypes.c_uint64 * 16)()
r._check(r._L.rt_debug_counters(r._h, a))
# word 8 holds a per-slot atomicMax of the wave lifetime; rt_debug_counters sums the 64 slots, so it
# is an upper bound of 64 x the longest wave: print the mean of the slot maxima
print(f"mean over counter slots of the longest wave lifetime: {a[8] / 64:.0f} clk")
a[8] = 0
names = ["(max lifetime)", "tile intake", "pop + primary rebuild", "lighting", "reflect scan+push", "(of intake: top_up)"]
tot = a[14]
print(f"{desc}: {a[15]} waves, mean wave lifetime {tot / max(a[15], 1):.0f} clk")
for i, nm in enumerate(names):
    print(f"  {nm:20s} {100.0 * a[8 + i] / max(tot, 1):6.2f} %")
print(f"  (of intake: vmcnt wait) {100.0 * a[7] / max(tot, 1):6.2f} %")
print(f"  {'other (setup/store)':20s} {100.0 * (tot - sum(a[8:14])) / max(tot, 1):6.2f} %")
if not os.environ.get("RT_CULL_STATS"):
    print(f"passes: {a[4]}, records per pass {a[5] / max(a[4], 1):.2f} of 64, passes run short with tiles left: {100.0 * a[6] / max(a[4], 1):.2f} %")
    sys.exit(0)
print(f"culling: {a[4] / max(a[5], 1):.2f} surviving spheres per lane and chunk, cull loop {a[7] / max(a[5], 1):.2f} turns, drain loop {a[6] / max(a[5], 1):.2f} turns per lane (= wave maxima)")
