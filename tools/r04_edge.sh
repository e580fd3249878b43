#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "noise or tnconv or edge or bitwise or graph_replay_is or S256 or fp8_configuration or denoise or merged_small" > gpurun_out/r04_edge_t.log 2>&1; rc=$?; tail -8 gpurun_out/r04_edge_t.log
[ $rc -ne 0 ] && exit $rc
for cfg in "64 128 100" "128 64 40" "256 32 30"; do set -- $cfg
  for i in 1 2; do
    timeout -k 10 200 python bench.py --size $1 --batch $2 --steps $3 --warmup 8 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_edge_s$1_new_$i.json 2>gpurun_out/r04_edge.err || { tail -5 gpurun_out/r04_edge.err; exit 1; }
    VG_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libvaegan_hip_prev.so timeout -k 10 200 python bench.py --size $1 --batch $2 --steps $3 --warmup 8 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_edge_s$1_old_$i.json 2>gpurun_out/r04_edge.err || { tail -5 gpurun_out/r04_edge.err; exit 1; }
    python - <<PY
import json
for t in ("old","new"):
    j=json.load(open("gpurun_out/r04_edge_s$1_%s_$i.json" % t)); print("S=$1", t, "run $i:", j["ms_per_step"], "ms; edge frac", j["roofline_edge"]["frac"], "avg us", j["roofline_edge"]["avg_launch_us"])
PY
  done
done
