#!/bin/bash
# round 4: the one-launch BatchNorm backward -- kernel tests, then the step with and without it (interleaved A/B)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "bn_backward or batchnorm" > gpurun_out/r04_op_t.log 2>&1; rc=$?; tail -15 gpurun_out/r04_op_t.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
  for m in 0 1; do
    VG_BN_ONEPASS=$m timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_op_bench_${m}_$i.json 2> gpurun_out/r04_op_bench_${m}_$i.err || exit 1
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_op_bench_${m}_$i.json")); print("ONEPASS=$m run $i:", j["ms_per_step"], "ms", j["kernel_launches_per_step"], "launches", "bn us/step", j["roofline_bn"]["us_per_step"], j["roofline_bn"]["launches_per_step"], j["losses"])
PY
  done
done
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "bitwise or benchmarked or teacher or all_parameter or graph_replay_is" > gpurun_out/r04_op_t2.log 2>&1; tail -5 gpurun_out/r04_op_t2.log
