#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "merged_small or step_prologue or bitwise or graph_replay_is or checkpoint or dropin or benchmarked" > gpurun_out/r04_merge_t.log 2>&1; rc=$?; tail -12 gpurun_out/r04_merge_t.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for m in 0 1; do
    VG_MERGE_SMALL=$m timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_merge_${m}_$i.json 2> gpurun_out/r04_merge_${m}_$i.err || exit 1
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_merge_${m}_$i.json")); print("MERGE=$m run $i:", j["ms_per_step"], "ms", j["kernel_launches_per_step"], "launches", j["losses"]["recon_loss"])
PY
  done
done
