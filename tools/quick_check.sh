#!/bin/bash
# Kernel + parity subsets and two bench lines in one GPU call (development loop).
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "${1:-bce or edge or adam or wgrad}" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_binding.py -x -q > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 300 --warmup 30 > gpurun_out/b$i.json 2> gpurun_out/b$i.err
  python -c "
import json; d = json.loads(open('gpurun_out/b$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
