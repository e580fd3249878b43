#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_k or few_row or conv2d_fprop or conv_transpose2d or linear_fused or patch_gather" > gpurun_out/r04_splitk_t.log 2>&1; rc=$?; tail -12 gpurun_out/r04_splitk_t.log
[ $rc -ne 0 ] && exit $rc
for cfg in "64 128 100" "128 64 40" "256 32 30"; do set -- $cfg
  for i in 1 2; do
  for m in 0 1; do
    VG_SPLITK_GENERAL=$m timeout -k 10 200 python bench.py --size $1 --batch $2 --steps $3 --warmup 8 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_splitk_s$1_${m}_$i.json 2>gpurun_out/r04_splitk_s$1_${m}_$i.err || { tail -5 gpurun_out/r04_splitk_s$1_${m}_$i.err; exit 1; }
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_splitk_s$1_${m}_$i.json")); print("S=$1 GENERAL=$m run $i:", j["ms_per_step"], "ms", j["kernel_launches_per_step"], "launches; gather-GEMM frac", j["roofline"]["frac"], j["roofline"]["launches_per_step"])
PY
  done; done
done
