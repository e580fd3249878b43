cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/r03_trace_s256b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_trace_s256b -- python3 bench.py --size 256 --batch 32 --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/r03_trace_s256b.log 2>&1 || exit 1
