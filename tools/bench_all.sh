#!/bin/bash
# Every workload DESIGN.md quotes, one JSON line each under gpurun_out/bench_<tag>_<workload>.json
# (run on the GPU box through gpurun): tools/bench_all.sh <tag>
tag=${1:-r3}
run() {  # name, bench args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" > gpurun_out/bench_${tag}_$name.json 2> gpurun_out/bench_${tag}_$name.err || { echo "$name failed"; tail -3 gpurun_out/bench_${tag}_$name.err; exit 1; }
  python - "$name" gpurun_out/bench_${tag}_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'{sys.argv[1]:12s} {d["value"]/1e3:8.1f} Grays/s  {d["ms_per_step"]:8.2f} ms/step  frames/step {d["config"]["frames_per_step"]}  '
      f'executed {r.get("frac")} algorithmic {r["algorithmic_frac"]:.2f} issue {r.get("issue_frac")}', flush=True)
PY
}
run c2 || exit 1
for wl in c1 t64 c5 refr refr40 cubes; do run $wl --workload $wl --no-target --no-cpu-baseline --no-single-frame --steps 8 --warmup 2 --batches 4 || exit 1; done
for wl in c3 cubes_glass; do run $wl --workload $wl --no-target --no-cpu-baseline --no-single-frame --steps 8 --warmup 2 --batches 1 || exit 1; done
run c4 --workload c4 --frames 16 --batches 4 --no-target --steps 5 --warmup 2 || exit 1
run c2_spp2 --samples 2 --frames 512 --batches 4 --no-target --no-cpu-baseline --no-single-frame --steps 8 --warmup 2 || exit 1
run c2_spp4 --samples 4 --frames 256 --batches 4 --no-target --no-cpu-baseline --no-single-frame --steps 8 --warmup 2 || exit 1
RT_BENCH_NO_TILE_FLUSH=1 run c2_noflush --no-target --no-cpu-baseline --no-single-frame --steps 8 --warmup 2 --batches 4 || exit 1
