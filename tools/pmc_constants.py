#!/usr/bin/env python3
"""Print bench.py's PMC table entry from a tools/profile.sh output directory (sums over the launch pair
rt_visibility_kernel + rt_stream_kernel, averages per dispatch).

    python tools/pmc_constants.py gpurun_out/prof_<tag> <workload> <profiles-name> <rays per launch pair>
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, wl, name, rays = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
tot = defaultdict(float)
cnt = defaultdict(int)
for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = "stream" if "rt_stream_kernel" in r["Kernel_Name"] else "vis" if "rt_visibility_kernel" in r["Kernel_Name"] else None
        if k:
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1


def g(c):
    return sum(tot[(k, c)] / cnt[(k, c)] for k in ("stream", "vis") if cnt[(k, c)])


flops = 64 * (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_TRANS_F32"))
hbm = 64 * g("TCC_EA0_WRREQ_sum") + 128 * g("TCC_EA0_RDREQ_sum")
print(f'    "{wl}": dict(profile="profiles/{name}_rt_{{stream,visibility}}_kernel.txt", flops={flops:.4g}, valu={g("SQ_INSTS_VALU"):.4g}, '
      f'salu={g("SQ_INSTS_SALU"):.4g}, hbm_bytes={hbm:.4g}, rays={rays:.4g}),')
