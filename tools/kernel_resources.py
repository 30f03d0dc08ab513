#!/usr/bin/env python3
"""Register / spill / scratch usage of every kernel of csrc/rt355.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage (device-only compile with the build's flags; needs no GPU).

    python tools/kernel_resources.py [-DNAME=VALUE ...] [--filter substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.build import HIPCC_FLAGS, SRC  # noqa: E402

defs = [a for a in sys.argv[1:] if a.startswith("-D")]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
cmd = ["hipcc", *flags, *defs, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", SRC,
       "-o", "/tmp/rt355_resources.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    key, _, val = m.group(1).partition(":")
    key, val = key.strip(), val.strip()
    if key == "Function Name":
        name = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
print(f"{'kernel':58s} {'sgpr':>5s} {'vgpr':>5s} {'s-spill':>7s} {'v-spill':>7s} {'scratch':>7s} {'occ':>4s} {'lds':>6s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:58]:58s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs', '?'):>5s} "
          f"{r.get('SGPRs Spill', '?'):>7s} {r.get('VGPRs Spill', '?'):>7s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>6s}")
