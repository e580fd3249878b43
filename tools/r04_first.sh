#!/bin/bash
# round 4, first GPU call: full GPU test suite, the default bench line, the launch sequence of one replayed step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t0.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_t0.log
tail -5 gpurun_out/r04_t0.log
timeout -k 10 300 python bench.py > gpurun_out/r04_bench0.json 2> gpurun_out/r04_bench0.err && cat gpurun_out/r04_bench0.json | head -c 1500
timeout -k 10 300 bash tools/graph_seq.sh > gpurun_out/r04_seq0.log 2>&1; cp gpurun_out/seq_step.txt gpurun_out/r04_seq0_step.txt; tail -60 gpurun_out/r04_seq0_step.txt
