#!/bin/bash
# The round's measurement set on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh r02
# bench line, rocprofv3 kernel stats of eager launches, two separate PMC passes (HBM-side bytes), then
# python profiles/summarize.py <tag>_bf16 ... and profiles/summarize_mfma.py <tag>_bf16 ... reduce them to profiles/<tag>_*.
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${TAG}_bench_bf16_default.json 2> gpurun_out/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_bf16_trace -- python3 bench.py --steps 10 --warmup 3 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bf16_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_bf16_fetch -- python3 bench.py --steps 3 --warmup 3 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bf16_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_bf16_write -- python3 bench.py --steps 3 --warmup 3 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bf16_write.log 2>&1
# MFMA utilisation as north_star words it: a PMC pass of its own (no trace domains beside --pmc on this pool)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_bf16_mfma -- python3 bench.py --steps 3 --warmup 3 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bf16_mfma.log 2>&1
# the reference's own arithmetic (exact-f32 MFMA): kernel stats + an MFMA-busy pass of its own (round-3 review item 3)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_fp32_trace -- python3 bench.py --dtype fp32 --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_fp32_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_fp32_mfma -- python3 bench.py --dtype fp32 --steps 2 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_fp32_mfma.log 2>&1
python3 bench.py --dtype fp32 --steps 20 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bench_fp32.json 2>/dev/null
# 512 images per GPU: the "rows per GPU cap the MFMA fraction" claim on this round's kernels
python3 bench.py --batch 512 --steps 20 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_bench_s64_b512.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_b512_trace -- python3 bench.py --batch 512 --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_b512_trace.log 2>&1
for cfg in "128 64" "256 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace_s$1 -- python3 bench.py --size $1 --batch $2 --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_trace_s$1.log 2>&1
done
python3 bench.py --size 128 --batch 64 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_s128.json 2> gpurun_out/${TAG}_s128.err
python3 bench.py --size 256 --batch 32 --no-cpu-baseline --no-extra-paths --steps 20 > gpurun_out/${TAG}_s256.json 2>/dev/null
python3 bench.py --dtype fp8 --size 256 --batch 32 --no-cpu-baseline --no-extra-paths --steps 20 > gpurun_out/${TAG}_s256_fp8.json 2>/dev/null
python3 bench.py --dtype fp8 --no-cpu-baseline --no-extra-paths > gpurun_out/${TAG}_s64_fp8.json 2>/dev/null
tail -c 600 gpurun_out/${TAG}_bench_bf16_default.json
