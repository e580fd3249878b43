#!/bin/bash
# Where does the side-graph penalty come from?  graph replay: serial / side graph / side graph replayed inline; eager: serial / side stream
set -e
mkdir -p gpurun_out
run() { # name, env..., extra args
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-paths $EXTRA > gpurun_out/ab_ov2_$name.json 2> gpurun_out/ab_ov2_$name.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_ov2_$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["ms_per_step"], flush=True)
PY
}
for rep in 1 2; do
  EXTRA="" run g0_$rep VG_OVERLAP=0
  EXTRA="" run g3_$rep VG_OVERLAP=3
  EXTRA="" run g3inline_$rep VG_OVERLAP=3 VG_SIDE_INLINE=1
  EXTRA="" run g1inline_$rep VG_OVERLAP=1 VG_SIDE_INLINE=1
  EXTRA="--graph 0" run e0_$rep VG_OVERLAP=0
  EXTRA="--graph 0" run e3_$rep VG_OVERLAP=3
done
