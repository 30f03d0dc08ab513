#!/bin/bash
# latency-related PMC pass: tools/pmc_lat.sh <workload> lib.so
wl=$1; lib=$2; root=$PWD
export TMPDIR=/tmp RT355_LIB=$root/$lib
out=$root/gpurun_out/pmc_lat; rm -rf "$out"; mkdir -p "$out"
i=0
for pmc in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target --workload $wl > "$out/pass$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "pass*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = "stream" if "stream" in r["Kernel_Name"] else "vis" if "visibility" in r["Kernel_Name"] else None
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]): print(f"   {c:28s} {tot[k][c]/cnt[k][c]:.4g}")
PY
