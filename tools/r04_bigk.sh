#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "few_row or failed_graph_capture or conv_transpose2d or linear_fused or bitwise" > gpurun_out/r04_bigk_t.log 2>&1; rc=$?; tail -8 gpurun_out/r04_bigk_t.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for m in 0 1; do
    VG_SPLITK_BIGK=$m timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_bigk_${m}_$i.json 2> gpurun_out/r04_bigk_${m}_$i.err || exit 1
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_bigk_${m}_$i.json")); print("BIGK=$m run $i:", j["ms_per_step"], "ms", j["kernel_launches_per_step"], "launches; gather-GEMM frac", j["roofline"]["frac"], "avg us", j["roofline"]["avg_launch_us"])
PY
  done
done
for sz in "128 64" "256 32"; do set -- $sz
  for m in 0 1; do
    VG_SPLITK_BIGK=$m timeout -k 10 200 python bench.py --size $1 --batch $2 --steps 40 --warmup 6 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_bigk_s$1_${m}.json 2>/dev/null || exit 1
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_bigk_s$1_${m}.json")); print("S=$1 BIGK=$m:", j["ms_per_step"], "ms; gather-GEMM frac", j["roofline"]["frac"])
PY
  done
done
