#!/bin/bash
# A/B of library builds: tools/ab.sh "<workloads>" lib1.so lib2.so ...   (one line per build and workload)
wls=$1; shift
for lib in "$@"; do
  for wl in $wls; do
    RT355_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-target --no-single-frame --batches ${AB_BATCHES:-1} --workload $wl ${AB_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$lib', '$wl', 'Mrays/s', round(d['value']), 'kern_ms', round(d['roofline']['kernel_ms'],3))
except Exception as e: print('$lib $wl ERR', l[-300:])
" || exit 1
  done
done
