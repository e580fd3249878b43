#!/bin/bash
# A/B: BatchNorm-backward sums in the data-gradient epilogue (VG_BNB=1, default) against the separate reduce pass (0).
set -e
mkdir -p gpurun_out
for rep in 1 2; do  # (VG_BNB defaults to 0)
  for m in 0 1; do
    VG_BNB=$m timeout -k 10 200 python bench.py --steps 300 --warmup 30 > gpurun_out/ab_bnb_${m}_${rep}.json 2> gpurun_out/ab_bnb_${m}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_bnb_${m}_${rep}.json").read().strip().splitlines()[-1])
print("VG_BNB=${m} rep ${rep}:", d["value"], d["ms_per_step"], flush=True)
PY
  done
done
