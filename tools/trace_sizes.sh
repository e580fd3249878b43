cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in "128 64" "256 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_trace_s$1 -- python3 bench.py --size $1 --batch $2 --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/r03_trace_s$1.log 2>&1 || exit 1
done
ls gpurun_out/r03_trace_s256/*/ | head
