#!/bin/bash
# tools/trace_only.sh <tag> [bench args]: rocprofv3 kernel trace + stats of bench.py (GPU box) -> gpurun_out/trace_<tag>/
tag=$1; shift
out=$PWD/gpurun_out/trace_$tag; mkdir -p "$out"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target $*"
cd /tmp && timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- $BENCH > "$out.log" 2>&1 || { echo "trace failed"; tail -3 "$out.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "*", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        print("  {Name:50.50s} calls={Calls} avg_ns={AverageNs} pct={Percentage}".format(**row))
PY
