#!/bin/bash
# round 4: gradient collectives captured inside the hipGraph -- tests + one-rank-over-RCCL bench against the plain step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r04_ddp_t.log 2>&1; rc=$?; tail -8 gpurun_out/r04_ddp_t.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for mode in plain inline cut; do
    case $mode in
      plain) envs="";;
      inline) envs="VAEGAN_FORCE_DIST=1 VAEGAN_DDP_CAPTURE=1";;
      cut) envs="VAEGAN_FORCE_DIST=1 VAEGAN_DDP_CAPTURE=0";;
    esac
    env $envs MASTER_PORT=$((29600 + RANDOM % 200)) timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-paths > gpurun_out/r04_ddp_${mode}_$i.json 2> gpurun_out/r04_ddp_${mode}_$i.err || { tail -5 gpurun_out/r04_ddp_${mode}_$i.err; exit 1; }
    python - <<PY
import json; j=json.load(open("gpurun_out/r04_ddp_${mode}_$i.json")); print("$mode run $i:", j["ms_per_step"], "ms", "segments", j["config"]["hip_graph_segments"], j["config"]["hip_graph"])
PY
  done
done
