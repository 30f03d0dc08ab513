#!/bin/bash
# tools/spp.sh "<workloads>" "<spp list>" lib...: ray rate per samples-per-pixel setting
wls=$1; spps=$2; shift 2
for lib in "$@"; do for wl in $wls; do for s in $spps; do
  RT355_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-target --workload $wl --samples $s | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$wl', 'spp', $s, round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
done; done; done
