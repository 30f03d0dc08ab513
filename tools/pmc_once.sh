#!/bin/bash
# tools/pmc_once.sh <tag> "<counters>" [bench args]: one rocprofv3 --pmc pass of bench.py (GPU box)
tag=$1; pmc=$2; shift 2
out=$PWD/gpurun_out/pmc_$tag; mkdir -p "$out"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target $*"
cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d "$out" -- $BENCH > "$out.log" 2>&1 || { echo "pmc pass failed"; tail -3 "$out.log"; exit 1; }
