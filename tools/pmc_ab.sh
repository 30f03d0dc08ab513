#!/bin/bash
# PMC comparison of library builds: tools/pmc_ab.sh <workload> lib1.so lib2.so ...  -> gpurun_out/pmc_ab/<lib>/passN
wl=$1; shift
root=$PWD
export TMPDIR=/tmp
for lib in "$@"; do
  export RT355_LIB=$root/$lib
  out=$root/gpurun_out/pmc_ab/$(basename $lib .so)_$wl
  rm -rf "$out"; mkdir -p "$out"
  i=0
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
    i=$((i+1))
    (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target --no-single-frame --batches ${AB_BATCHES:-1} --workload $wl > "$out/pass$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
  done
done
python3 - "$root/gpurun_out/pmc_ab" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*"))):
    tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "pass*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = "stream" if "stream" in r["Kernel_Name"] else "vis" if "visibility" in r["Kernel_Name"] else None
            if k:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in tot:
        print(os.path.basename(d), k, " ".join(f"{c}={tot[k][c]/cnt[k][c]:.4g}" for c in sorted(tot[k])))
PY
