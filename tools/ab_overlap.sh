#!/bin/bash
# A/B: side-stream schedule (VG_OVERLAP bit mask: 1 Generator Adam + re-pack beside the Encoder's backward,
# 2 dead Discriminator weight gradients beside it too) against the serial schedule (0), interleaved.
set -e
mkdir -p gpurun_out
MODES=${MODES:-"0 1 3"}
for rep in 1 2; do
  for m in $MODES; do
    VG_OVERLAP=$m timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-paths > gpurun_out/ab_ov_${m}_${rep}.json 2> gpurun_out/ab_ov_${m}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_ov_${m}_${rep}.json").read().strip().splitlines()[-1])
print("VG_OVERLAP=${m} rep ${rep}:", d["value"], d["ms_per_step"], flush=True)
PY
  done
done
