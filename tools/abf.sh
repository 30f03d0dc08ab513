#!/bin/bash
# tools/abf.sh "<workloads>" "<lib:frames> ..."  — A/B with a frames-per-step setting per library
wls=$1; shift
for lf in "$@"; do lib=${lf%%:*}; fr=${lf##*:}
  for wl in $wls; do
    RT355_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-target --workload $wl --frames $fr 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('$lib', 'frames=$fr', '$wl', 'Mrays/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
  done
done
