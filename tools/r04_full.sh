#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_full_t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_full_t.log; tail -6 gpurun_out/r04_full_t.log
