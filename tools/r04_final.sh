#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_final_t.log; tail -4 gpurun_out/r04_final_t.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final_smoke.log 2>&1; tail -2 gpurun_out/r04_final_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; python - <<'PY'
import json; j=json.load(open("gpurun_out/r04_final_bench.json")); print(j["value"], j["ms_per_step"], j["kernel_launches_per_step"], j["roofline"]["frac"], j["roofline_edge"]["frac"], j["roofline_bn"]["frac"], j["parity_path"]["roofline"]["frac"], j["cpu_baseline"]["value"], j["dropin_path"]["graphed"]["value"])
PY
