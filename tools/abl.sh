#!/bin/bash
# timing of ablation builds (results are wrong by construction): kernel ms per launch and rays counted
wls=$1; shift
for lib in "$@"; do for wl in $wls; do
  RT355_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-target --workload $wl 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('$lib', '$wl', 'kern_ms', round(d['roofline']['kernel_ms'],3), 'rays/frame', round(d['rays_per_frame']))"
done; done
