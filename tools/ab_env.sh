#!/bin/bash
# Interleaved A/B of bench.py under values of one environment variable:  tools/ab_env.sh VAR "v1 v2 ..." [reps=2] [bench args]
set -e
VAR=$1; VALS=$2; REPS=${3:-2}; shift 3 || true
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-paths "$@" > gpurun_out/ab_env_${VAR}_$(basename $v)_${rep}.json 2> gpurun_out/ab_env_${VAR}_$(basename $v)_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_env_${VAR}_$(basename $v)_${rep}.json").read().strip().splitlines()[-1])
print("$VAR=$v rep $rep:", d["value"], d["ms_per_step"], "gg frac", d["roofline"]["frac"], flush=True)
PY
  done
done
