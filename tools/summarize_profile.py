#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into one text file for profiles/.

    python tools/summarize_profile.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "rt_render_kernel"
print(f"# rocprofv3 summary of {d} (kernel filter: {kern})")
for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
    print("\n## kernel stats (rocprofv3 --kernel-trace --stats)")
    for row in csv.DictReader(open(f)):
        print("  {Name:60.60s} calls={Calls} total_ns={TotalDurationNs} avg_ns={AverageNs} min={MinNs} max={MaxNs} pct={Percentage}".format(**row))
for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    if rows:
        r = rows[-1]
        keys = [k for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                            "Workgroup_Size", "Grid_Size") if k in r]
        print("\n## last dispatch:", {k: r[k] for k in keys})
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        print(f"  {len(dur)} dispatches, avg {sum(dur)/len(dur)/1e3:.1f} us, min {min(dur)/1e3:.1f}, max {max(dur)/1e3:.1f}")
tot = defaultdict(float)
cnt = defaultdict(int)
for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
if tot:
    print("\n## PMC counters, average per dispatch of", kern)
    avg = {k: tot[k] / cnt[k] for k in tot}
    for k in sorted(avg):
        print(f"  {k:28s} {avg[k]:18.1f}   ({cnt[k]} dispatches)")
    g = avg.get
    print("\n## derived")
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print(f"  active-lane ratio  SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU) = {g('SQ_THREAD_CYCLES_VALU')/(64*g('SQ_ACTIVE_INST_VALU')):.3f}")
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        print(f"  VALU busy          SQ_ACTIVE_INST_VALU*4/SQ_BUSY_CYCLES (per SE-summed busy) = {4*g('SQ_ACTIVE_INST_VALU')/g('SQ_BUSY_CYCLES'):.3f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        print(f"  VALU share of wave cycles  = {g('SQ_ACTIVE_INST_VALU')/g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print(f"  VALU instructions per wave = {g('SQ_INSTS_VALU')/g('SQ_WAVES'):.0f}   SALU per wave = {g('SQ_INSTS_SALU',0)/g('SQ_WAVES'):.0f}")
    fl = g("SQ_INSTS_VALU_ADD_F32", 0) + g("SQ_INSTS_VALU_MUL_F32", 0) + 2 * g("SQ_INSTS_VALU_FMA_F32", 0) + g("SQ_INSTS_VALU_TRANS_F32", 0)
    if fl:
        print(f"  executed FP32 wave-instructions: add {g('SQ_INSTS_VALU_ADD_F32',0):.0f} mul {g('SQ_INSTS_VALU_MUL_F32',0):.0f} "
              f"fma {g('SQ_INSTS_VALU_FMA_F32',0):.0f} trans {g('SQ_INSTS_VALU_TRANS_F32',0):.0f}")
        print(f"  executed flops per dispatch (x64 lanes, full-wave upper bound) = {fl*64:.3e}")
    if g("TCC_EA0_WRREQ_sum") is not None:
        print(f"  HBM write bytes ~ TCC_EA0_WRREQ*64 = {g('TCC_EA0_WRREQ_sum',0)*64:.3e}  read ~ RDREQ*64(x2 see guide) = {g('TCC_EA0_RDREQ_sum',0)*64:.3e}")
