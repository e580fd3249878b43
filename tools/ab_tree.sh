#!/bin/bash
# Interleaved same-box A/B of bench.py between this tree and a second checkout (built beforehand):
#   tools/ab_tree.sh scratch/base [reps=3] [bench args]
B=$1; REPS=${2:-3}; shift 2 || true
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
  for t in "$B" .; do
    ( cd $t && timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-paths "$@" ) > gpurun_out/ab_tree_$(basename $t)_${rep}.json 2> gpurun_out/ab_tree_$(basename $t)_${rep}.err || { echo "bench failed in $t"; tail -3 gpurun_out/ab_tree_$(basename $t)_${rep}.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_tree_$(basename $t)_${rep}.json").read().strip().splitlines()[-1])
print("$t rep $rep:", d["value"], d["ms_per_step"], "gg frac", d["roofline"]["frac"], flush=True)
PY
  done
done
