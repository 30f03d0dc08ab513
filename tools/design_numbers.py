#!/usr/bin/env python3
"""Rewrites the round-3 results table of DESIGN.md section 6 (between the r3-table markers) from profiles/bench_r3z/*.json.
    python tools/design_numbers.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rec(wl):
    return json.loads(open(os.path.join(ROOT, "profiles", "bench_r3z", f"bench_r3z_{wl}.json")).read().strip().splitlines()[-1])


rows = [("config 2 (1080p, 8 spheres, D5) — the bench line", "c2", 619, 359),
        ("north-star target scene (1080p, 64 spheres, D5)", "t64", 556, 266),
        ("config 1 (640×480, default scene, D3)", "c1", 559, None),
        ("config 3 (4K, 64 spheres, D8)", "c3", 612, None),
        ("config 4 on one GPU (8K, 1024 spheres, D5)", "c4", 340, 76),
        ("config 5 (256-frame orbit, 1080p) on one GPU", "c5", 602, None),
        ("§8f: `refractive_scene()` 1080p D5", "refr", 356, 194),
        ("§8f: 40 spheres, every third of glass", "refr40", 331, 85),
        ("§8f: `cube_scene()` 1080p D5", "cubes", 335, 149),
        ("§8f: `cube_scene(glass=True)` 1080p D5", "cubes_glass", 220, 114)]
t = "| workload | Grays/s round 3 | round 2 | round 1 |\n|---|---|---|---|\n"
for name, wl, r2, r1 in rows:
    t += f"| {name} | **{rec(wl)['value'] / 1e3:.0f}** | {r2 or '—'} | {r1 or '—'} |\n"
t += f"| config 2 at 2 / 4 samples per pixel | {rec('c2_spp2')['value'] / 1e3:.0f} / {rec('c2_spp4')['value'] / 1e3:.0f} | 575 / 600 | — |\n"
t += f"| config 2 with `RT_OPT_TILE_FLUSH` 0 (pixels stored one by one) | {rec('c2_noflush')['value'] / 1e3:.0f} | 596 | — |\n"
c2 = rec("c2")
r = c2["roofline"]
t += f"""
Config 2's launch pair: {r['kernel_ms']:.2f} ms per 1024 frames (`profiles/r3z_c2_*` has the two kernels apart), executed FP32
{r['frac']:.3f} of peak, issue slots {r['issue_frac']:.2f} (vector alone {r['valu_issue_frac']:.2f}), HBM {r['traffic']:.3g} B per launch pair for
{r['frame_bytes_per_launch']:.3g} B of frames ({r['traffic'] / r['frame_bytes_per_launch']:.2f}×; round 2: 1.78×).  Single frame through the drop-in call: {c2['single_frame']['ms_per_call']:.2f} ms
({c2['single_frame']['mrays_per_s'] / 1e3:.0f} Grays/s, bound by the copy of the frame to the host).  CPU baseline (FP64 port, {c2['cpu_baseline']['cores']} cores):
{c2['cpu_baseline']['value']:.0f} Mrays/s; the 64-sphere target on the same line: {c2['target_64sph']['value'] / 1e3:.0f} Grays/s.
"""
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- r3-table-begin -->"), s.index("<!-- r3-table-end -->")
s = s[:a] + "<!-- r3-table-begin -->\n" + t + s[b:]
open(p, "w").write(s)
print(t)
